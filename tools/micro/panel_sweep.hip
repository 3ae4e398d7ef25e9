// Micro-benchmark: the sweep of the delayed-update kernels (stream3_kernel / dshard_kernel flush_pending) as a PANEL sweep:
// the K pending normalised pivot rows are staged in LDS one column panel at a time and every row of the workgroup gets its
// K eliminations from there -- x[r][c] = (..((x - coef[0][r] * p[0][c]) - coef[1][r] * p[1][c]) ..), each product and
// difference rounded on its own -- instead of re-reading the pending rows from L2 for every pair of rows (round 2:
// 16385^2 sweeps at 4.6 TB/s with 8 pending, the bare sweep at 6.1).  The design input for the round-3 flush.
//   PU   16-byte units of a row per panel (panel = 2 PU columns = 16 PU bytes of every row)
//   T    lanes per workgroup; T / PU rows side by side when PU < T, PU / T units per lane when PU > T
//   D    row slots in flight per lane (loads of the next batch are issued before the current one is computed: PIPE)
//   K    pending pivots
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o panel_sweep panel_sweep.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double v2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ v2 ld(const double *p) {
    const v2 *q = reinterpret_cast<const v2 *>(p);
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void st(double *p, v2 v) {
    v2 *q = reinterpret_cast<v2 *>(p);
    if (NT)
        __builtin_nontemporal_store(v, q);
    else
        *q = v;
}

// rows of a workgroup: b, b + NB, ...; row slot s of batch i = row index i * RS * D + d * RS + sub (sub = tid / PU when PU < T)
template <int T, int PU, int D, int K, bool NT, bool PIPE>
__global__ __launch_bounds__(T) void panel_sweep(double *__restrict__ a, const double *__restrict__ pend, const double *__restrict__ coef,
                                                 int rows, int pitch) {
    constexpr int RS = PU < T ? T / PU : 1; // rows side by side
    constexpr int U = PU > T ? PU / T : 1;  // units per lane and row
    constexpr int LU = PU < T ? PU : T;     // lanes across a row segment
    extern __shared__ __attribute__((aligned(16))) double lds[]; // [K][2 PU] pending panel, then [K][rpw] coefficients
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x;
    const int sub = tid / LU, lane = tid % LU;
    const int count = b < rows ? (rows - 1 - b) / NB + 1 : 0;
    const int rpw = (rows + NB - 1) / NB;
    double *cf = lds + (size_t)K * 2 * PU;
    for (int i = tid; i < K * rpw; i += T) {
        const int k = i / rpw, r = i % rpw;
        cf[i] = r < count ? coef[(size_t)k * rows + b + r * NB] : 0.0;
    }
    const int panels = pitch / (2 * PU);
    for (int pnl = 0; pnl < panels; pnl++) {
        const int c0 = pnl * 2 * PU;
        __syncthreads(); // (the previous panel's readers are through)
        for (int i = tid; i < K * PU; i += T) {
            const int k = i / PU, u = i % PU;
            *reinterpret_cast<v2 *>(lds + (size_t)k * 2 * PU + 2 * u) = *reinterpret_cast<const v2 *>(pend + (size_t)k * pitch + c0 + 2 * u);
        }
        __syncthreads();
        const int per_batch = RS * D;
        v2 x[D][U], nx[D][U];
        auto load = [&](int i0, v2 (&dst)[D][U]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int ri = i0 + d * RS + sub;
                const int r = b + (ri < count ? ri : 0) * NB;
                const double *src = a + (size_t)r * pitch + c0;
#pragma unroll
                for (int u = 0; u < U; u++) dst[d][u] = ld<NT>(src + 2 * (lane + u * LU));
            }
        };
        if (PIPE && count > 0) load(0, x);
        for (int i0 = 0; i0 < count; i0 += per_batch) {
            if (!PIPE) load(i0, x);
            if (PIPE && i0 + per_batch < count) load(i0 + per_batch, nx);
#pragma unroll 1
            for (int k = 0; k < K; k++) { // (a run-time loop, as in the kernels: the number of pending pivots varies)
                v2 pn[U];
#pragma unroll
                for (int u = 0; u < U; u++) pn[u] = *reinterpret_cast<const v2 *>(lds + (size_t)k * 2 * PU + 2 * (lane + u * LU));
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const int ri = i0 + d * RS + sub;
                    const double c = cf[k * rpw + (ri < count ? ri : 0)];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const double px = c * pn[u].x, py = c * pn[u].y;
                        x[d][u].x = x[d][u].x - px;
                        x[d][u].y = x[d][u].y - py;
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int ri = i0 + d * RS + sub;
                if (ri < count) {
                    double *dst = a + (size_t)(b + ri * NB) * pitch + c0;
#pragma unroll
                    for (int u = 0; u < U; u++) st<NT>(dst + 2 * (lane + u * LU), x[d][u]);
                }
            }
            if (PIPE) {
#pragma unroll
                for (int d = 0; d < D; d++)
#pragma unroll
                    for (int u = 0; u < U; u++) x[d][u] = nx[d][u];
            }
        }
    }
}

template <int T, int PU, int D, int K, bool NT, bool PIPE>
void run(double *a, double *pend, double *coef, int rows, int pitch) {
    if (pitch % (2 * PU)) return;
    const int rpw = (rows + 255) / 256;
    const size_t shmem = sizeof(double) * ((size_t)K * 2 * PU + (size_t)K * rpw);
    if (shmem > 158 * 1024) return;
    auto fn = &panel_sweep<T, PU, D, K, NT, PIPE>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 5;
    for (int i = 0; i < reps + 1; i++) {
        if (i == 1) hipEventRecord(e0);
        fn<<<256, T, shmem>>>(a, pend, coef, rows, pitch);
    }
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) {
        printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
        return;
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * rows * (double)pitch * 8;
    printf("%6dx%-6d panel %5d cols (%5d B/row) T=%d rows-side-by-side=%d D=%2d K=%2d %s %s LDS %6zu B  %9.1f us  %.2f TB/s\n", rows, pitch, 2 * PU,
           16 * PU, T, PU < T ? T / PU : 1, D, K, NT ? "nt   " : "plain", PIPE ? "pipelined" : "unpiped  ", shmem, 1e3 * ms / reps,
           bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const int shapes[][2] = {{16385, 16384}, {2049, 16384}, {8193, 8192}, {4097, 4096}};
    for (auto &s : shapes) {
        const int rows = s[0], pitch = s[1];
        const size_t bytes = (size_t)rows * pitch * 8;
        double *a, *p, *coef;
        if (hipMalloc(&a, bytes) != hipSuccess) return 1;
        hipMalloc(&p, 16 * (size_t)pitch * 8);
        hipMalloc(&coef, 16 * (size_t)rows * 8);
        hipMemset(a, 0, bytes);
        hipMemset(p, 0, 16 * (size_t)pitch * 8);
        hipMemset(coef, 0, 16 * (size_t)rows * 8);
        const bool big = bytes > (200u << 20);
        // K = 8: 2 KB / 4 KB / 8 KB / 16 KB of a row per panel
#define BOTH(T, PU, D, K)                                           \
    if (big) {                                                      \
        run<T, PU, D, K, true, true>(a, p, coef, rows, pitch);      \
        run<T, PU, D, K, true, false>(a, p, coef, rows, pitch);     \
    } else {                                                        \
        run<T, PU, D, K, false, true>(a, p, coef, rows, pitch);     \
        run<T, PU, D, K, true, true>(a, p, coef, rows, pitch);      \
    }
        BOTH(512, 128, 4, 8)
        BOTH(512, 128, 8, 8)
        BOTH(512, 256, 4, 8)
        BOTH(512, 256, 8, 8)
        BOTH(512, 512, 4, 8)
        BOTH(512, 512, 8, 8)
        BOTH(512, 512, 16, 8)
        BOTH(512, 1024, 4, 8)
        BOTH(512, 1024, 8, 8)
        BOTH(1024, 1024, 4, 8)
        BOTH(1024, 1024, 8, 8)
        // K = 16 (twice the depth in the same LDS: panels half as wide)
        BOTH(512, 256, 8, 16)
        BOTH(512, 512, 4, 16)
        BOTH(512, 512, 8, 16)
        BOTH(1024, 512, 4, 16)
        // K = 4
        BOTH(512, 512, 8, 4)
        BOTH(512, 1024, 8, 4)
        hipFree(a);
        hipFree(p);
        hipFree(coef);
    }
    return 0;
}
