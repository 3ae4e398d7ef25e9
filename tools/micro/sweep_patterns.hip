// Micro-benchmark: which shape of an IN-PLACE rank-1 sweep x[r][c] -= coef[r] * p[c] streams fastest from HBM on MI355X
// (the design input for sweep_kernel.cuh).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, run on the GPU box.
//   variants: T lanes per workgroup, J 16-byte units per lane and row (T * J * 2 = pitch), D rows in flight per lane,
//             cache policy of the row loads / stores (plain | nt), rows of a workgroup strided (b, b + NB, ...) or in a block,
//             workgroups per CU (grid = 256 * G).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double v2 __attribute__((ext_vector_type(2)));

template <int T, int J, int D, bool NT, bool BLOCKED>
__global__ __launch_bounds__(T) void sweep(double *__restrict__ a, const double *__restrict__ p, const double *__restrict__ coef,
                                           int rows, int pitch) {
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x;
    v2 pr[J];
#pragma unroll
    for (int j = 0; j < J; j++) pr[j] = *reinterpret_cast<const v2 *>(p + 2 * (tid + j * T));
    const int per = (rows + NB - 1) / NB;
    const int r0 = BLOCKED ? b * per : b, step = BLOCKED ? 1 : NB;
    const int count = BLOCKED ? (r0 + per <= rows ? per : (rows > r0 ? rows - r0 : 0)) : (b < rows ? (rows - 1 - b) / NB + 1 : 0);
    for (int i = 0; i < count; i += D) {
        v2 x[D][J];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int r = r0 + (i + d < count ? i + d : i) * step;
            const double *src = a + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const v2 *q = reinterpret_cast<const v2 *>(src + 2 * (tid + j * T));
                x[d][j] = NT ? __builtin_nontemporal_load(q) : *q;
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (i + d >= count) break;
            const int r = r0 + (i + d) * step;
            const double c = coef[r];
            double *dst = a + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                v2 v = x[d][j];
                const double px = c * pr[j].x, py = c * pr[j].y;
                v.x = v.x - px;
                v.y = v.y - py;
                v2 *q = reinterpret_cast<v2 *>(dst + 2 * (tid + j * T));
                if (NT)
                    __builtin_nontemporal_store(v, q);
                else
                    *q = v;
            }
        }
    }
}

// the same with the pivot row in LDS instead of registers (what frees 4 J registers per lane for a second row in flight)
template <int T, int J, int D, bool NT>
__global__ __launch_bounds__(T) void sweep_lds(double *__restrict__ a, const double *__restrict__ p, const double *__restrict__ coef,
                                               int rows, int pitch) {
    extern __shared__ double prow[];
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x;
#pragma unroll
    for (int j = 0; j < J; j++) *reinterpret_cast<v2 *>(prow + 2 * (tid + j * T)) = *reinterpret_cast<const v2 *>(p + 2 * (tid + j * T));
    __syncthreads();
    const int count = b < rows ? (rows - 1 - b) / NB + 1 : 0;
    for (int i = 0; i < count; i += D) {
        v2 x[D][J];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int r = b + (i + d < count ? i + d : i) * NB;
            const double *src = a + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const v2 *q = reinterpret_cast<const v2 *>(src + 2 * (tid + j * T));
                x[d][j] = NT ? __builtin_nontemporal_load(q) : *q;
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (i + d >= count) break;
            const int r = b + (i + d) * NB;
            const double c = coef[r];
            double *dst = a + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                v2 v = x[d][j];
                const v2 pr = *reinterpret_cast<const v2 *>(prow + 2 * (tid + j * T));
                const double px = c * pr.x, py = c * pr.y;
                v.x = v.x - px;
                v.y = v.y - py;
                v2 *q = reinterpret_cast<v2 *>(dst + 2 * (tid + j * T));
                if (NT)
                    __builtin_nontemporal_store(v, q);
                else
                    *q = v;
            }
        }
    }
}
template <int T, int J, int D, bool NT>
void run_lds(double *a, double *p, double *coef, int rows, int pitch) {
    if (T * J * 2 != pitch) return;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_lds<T, J, D, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, pitch * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 6;
    for (int i = 0; i < reps + 2; i++) {
        if (i == 2) hipEventRecord(e0);
        sweep_lds<T, J, D, NT><<<256, T, pitch * 8>>>(a, p, coef, rows, pitch);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * rows * (double)pitch * 8;
    printf("%6dx%-6d T=%4d J=%d D=%d %s strided G=1 prow-in-LDS %9.1f us  %.2f TB/s\n", rows, pitch, T, J, D, NT ? "nt   " : "plain", 1e3 * ms / reps,
           bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
}

template <int T, int J, int D, bool NT, bool BLOCKED>
void run(const char *name, double *a, double *p, double *coef, int rows, int pitch, int G) {
    if (T * J * 2 != pitch) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 6;
    for (int i = 0; i < reps + 2; i++) {
        if (i == 2) hipEventRecord(e0);
        sweep<T, J, D, NT, BLOCKED><<<256 * G, T>>>(a, p, coef, rows, pitch);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * rows * (double)pitch * 8;
    printf("%6dx%-6d T=%4d J=%d D=%d %s %s G=%d %-10s %9.1f us  %.2f TB/s\n", rows, pitch, T, J, D, NT ? "nt   " : "plain", BLOCKED ? "blocked" : "strided",
           G, name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const int shapes[][2] = {{16385, 16384}, {2049, 16384}, {8193, 8192}, {4097, 16384}};
    for (auto &s : shapes) {
        const int rows = s[0], pitch = s[1];
        const size_t bytes = (size_t)rows * pitch * 8;
        double *a, *p, *coef;
        if (hipMalloc(&a, bytes) != hipSuccess) return 1;
        hipMalloc(&p, pitch * 8);
        hipMalloc(&coef, rows * 8);
        hipMemset(a, 0, bytes);
        hipMemset(p, 0, pitch * 8);
        hipMemset(coef, 0, rows * 8);
        // pitch 16384: T*J = 8192
        run<1024, 8, 1, false, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 8, 2, false, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 8, 1, true, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 8, 2, true, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 8, 2, true, true>("", a, p, coef, rows, pitch, 1);
        run<1024, 8, 1, true, false>("", a, p, coef, rows, pitch, 2);
        run<512, 16, 1, true, false>("", a, p, coef, rows, pitch, 1);
        run<512, 16, 2, true, false>("", a, p, coef, rows, pitch, 1);
        run<512, 16, 2, false, false>("", a, p, coef, rows, pitch, 1);
        run_lds<1024, 8, 2, true>(a, p, coef, rows, pitch);
        run_lds<1024, 8, 2, false>(a, p, coef, rows, pitch);
        run_lds<1024, 8, 3, true>(a, p, coef, rows, pitch);
        run_lds<1024, 4, 2, true>(a, p, coef, rows, pitch);
        run_lds<1024, 4, 3, true>(a, p, coef, rows, pitch);
        run<512, 16, 1, true, false>("", a, p, coef, rows, pitch, 2);
        run<512, 16, 1, false, false>("", a, p, coef, rows, pitch, 2);
        run<256, 32, 1, true, false>("", a, p, coef, rows, pitch, 4);
        // pitch 8192: T*J = 4096
        run<1024, 4, 1, false, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 4, 2, false, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 4, 2, true, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 4, 4, true, false>("", a, p, coef, rows, pitch, 1);
        run<1024, 4, 2, true, true>("", a, p, coef, rows, pitch, 1);
        run<512, 8, 2, true, false>("", a, p, coef, rows, pitch, 2);
        run<256, 16, 2, true, false>("", a, p, coef, rows, pitch, 4);
        hipFree(a);
        hipFree(p);
        hipFree(coef);
    }
    return 0;
}
