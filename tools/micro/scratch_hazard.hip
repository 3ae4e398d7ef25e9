// Reduced case for DESIGN.md 4.8 ("kernels that spill computed wrong rows"): is a VGPR that a store is still reading
// protected against the VALU instruction that overwrites it right behind the store, on gfx950?
// Each variant stores a register (pattern A), overwrites that register in the very next instruction (pattern B), waits,
// loads the location back and counts the lanes that read anything but A.  Variants:
//   scratch1 / scratch2 / scratch4     scratch_store_dword / x2 / x4  off, v, off          (what hipcc emits for spills)
//   scratch4_s                         scratch_store_dwordx4 off, v, s[off]                 (SGPR offset)
//   scratch4_f64                       ... overwritten by v_add_f64 instead of v_mov_b32    (the elimination's own instruction)
//   buffer4_s (control)                buffer_store_dwordx4 v, v, s[rsrc], s[soffset != 0]  (the hazard of DESIGN.md 4.7)
//   buffer4_0 (control)                the same with soffset = 0
// hipcc --offload-arch=gfx950 -O3 -o scratch_hazard scratch_hazard.hip ; run on the GPU box (prints one line per variant).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            return 1;                                                              \
        }                                                                          \
    } while (0)

enum { SCRATCH1, SCRATCH2, SCRATCH4, SCRATCH4_S, SCRATCH4_F64, BUFFER4_S, BUFFER4_0, NVAR };
static const char *kNames[NVAR] = {"scratch_store_dword  + v_mov", "scratch_store_dwordx2 + v_mov", "scratch_store_dwordx4 + v_mov",
                                   "scratch_store_dwordx4 saddr + v_mov", "scratch_store_dwordx4 + v_add_f64",
                                   "buffer_store_dwordx4 soffset!=0 + v_mov (control)", "buffer_store_dwordx4 soffset=0 + v_mov (control)"};

template <int V>
__global__ __launch_bounds__(512) void hazard(unsigned long long *bad, unsigned *buf, int iters) {
    volatile unsigned priv[64]; // forces a private segment: the asm below addresses its first 32 bytes
    priv[threadIdx.x & 63] = 0;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long wrong = 0;
    unsigned *mine = buf + (size_t)tid * 8; // 32 bytes per lane (buffer variants)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
    for (int it = 0; it < iters; it++) {
        const unsigned a = tid * 2654435761u + it * 40503u + 1u, b = ~a;
        unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if constexpr (V == SCRATCH1) {
            asm volatile("v_mov_b32 v40, %1\n\t"
                         "scratch_store_dword off, v40, off offset:0\n\t"
                         "v_mov_b32 v40, %2\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "scratch_load_dword %0, off, off offset:0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=v"(r0) : "v"(a), "v"(b) : "v40", "memory");
            wrong += r0 != a;
        } else if constexpr (V == SCRATCH2) {
            asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %2\n\t"
                         "scratch_store_dwordx2 off, v[40:41], off offset:0\n\t"
                         "v_mov_b32 v40, %3\n\tv_mov_b32 v41, %3\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "scratch_load_dwordx2 v[42:43], off, off offset:0\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "v_mov_b32 %0, v42\n\tv_mov_b32 %1, v43"
                         : "=v"(r0), "=v"(r1) : "v"(a), "v"(b) : "v40", "v41", "v42", "v43", "memory");
            wrong += (r0 != a) + (r1 != a);
        } else if constexpr (V == SCRATCH4 || V == SCRATCH4_S || V == SCRATCH4_F64) {
            if constexpr (V == SCRATCH4)
                asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                             "scratch_store_dwordx4 off, v[40:43], off offset:0\n\t"
                             "v_mov_b32 v43, %5\n\tv_mov_b32 v42, %5\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v40, %5\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "scratch_load_dwordx4 v[44:47], off, off offset:0\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a), "v"(b)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            else if constexpr (V == SCRATCH4_S) {
                const int soff = 16;
                asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                             "scratch_store_dwordx4 off, v[40:43], %6 offset:0\n\t"
                             "v_mov_b32 v43, %5\n\tv_mov_b32 v42, %5\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v40, %5\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "scratch_load_dwordx4 v[44:47], off, %6 offset:0\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a), "v"(b), "s"(soff)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            } else
                asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                             "v_mov_b32 v48, %5\n\tv_mov_b32 v49, %5\n\t"
                             "scratch_store_dwordx4 off, v[40:43], off offset:0\n\t"
                             "v_add_f64 v[42:43], v[48:49], v[48:49]\n\t"
                             "v_add_f64 v[40:41], v[48:49], v[48:49]\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "scratch_load_dwordx4 v[44:47], off, off offset:0\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a), "v"(b)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "memory");
            wrong += (r0 != a) + (r1 != a) + (r2 != a) + (r3 != a);
        } else {
            // lane offset in a VGPR; the SGPR offset carries 16 bytes (BUFFER4_S) or nothing (BUFFER4_0)
            const unsigned voff = tid * 32u + (V == BUFFER4_S ? 0u : 16u);
            const int soff = V == BUFFER4_S ? 16 : 0;
            asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %4\n\t"
                         "buffer_store_dwordx4 v[40:43], %6, %7, %8 offen\n\t"
                         "v_mov_b32 v43, %5\n\tv_mov_b32 v42, %5\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v40, %5\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "buffer_load_dwordx4 v[44:47], %6, %7, %8 offen\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a), "v"(b), "v"(voff), "s"(rsrc), "s"(soff)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            wrong += (r0 != a) + (r1 != a) + (r2 != a) + (r3 != a);
        }
    }
    if (mine[0] == 0xdeadbeefu && priv[1] == 77u) wrong += 1; // (keeps both allocations alive)
    if (wrong) atomicAdd(bad, wrong);
}

template <int V>
int run(unsigned long long *bad, unsigned *buf, int blocks, int iters) {
    CHECK(hipMemset(bad, 0, 8));
    hazard<V><<<blocks, 512>>>(bad, buf, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    const double words = (double)blocks * 512 * iters * (V == SCRATCH1 ? 1 : V == SCRATCH2 ? 2 : 4);
    printf("%-55s %12llu wrong words of %.3g\n", kNames[V], h, words);
    fflush(stdout);
    return 0;
}

int main() {
    const int blocks = 1024, iters = 20000;
    unsigned long long *bad;
    unsigned *buf;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&buf, (size_t)blocks * 512 * 32 + 64));
    CHECK(hipMemset(buf, 0, (size_t)blocks * 512 * 32 + 64));
    if (run<SCRATCH1>(bad, buf, blocks, iters)) return 1;
    if (run<SCRATCH2>(bad, buf, blocks, iters)) return 1;
    if (run<SCRATCH4>(bad, buf, blocks, iters)) return 1;
    if (run<SCRATCH4_S>(bad, buf, blocks, iters)) return 1;
    if (run<SCRATCH4_F64>(bad, buf, blocks, iters)) return 1;
    if (run<BUFFER4_S>(bad, buf, blocks, iters)) return 1;
    if (run<BUFFER4_0>(bad, buf, blocks, iters)) return 1;
    return 0;
}
