// Micro-benchmark behind DESIGN.md's "in-place vs ping-pong" note: x = x - c * p over a tableau-sized
// array, in place and out of place, at several sizes (does the 256 MB Infinity Cache make an in-place
// sweep of a 134 MB tableau faster than the ping-pong sweep the streaming kernels do today?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(1024) void sweep(const double *__restrict__ a, double *__restrict__ b, const double *__restrict__ p,
                                              size_t rows, size_t pitch) {
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const double *src = a + r * pitch;
        double *dst = b + r * pitch;
        for (size_t c = 2 * threadIdx.x; c < pitch; c += 2 * blockDim.x) {
            v2 x = *reinterpret_cast<const v2 *>(src + c);
            const v2 q = *reinterpret_cast<const v2 *>(p + c);
            x.x = x.x - 0.5 * q.x;
            x.y = x.y - 0.5 * q.y;
            if (NT)
                __builtin_nontemporal_store(x, reinterpret_cast<v2 *>(dst + c));
            else
                *reinterpret_cast<v2 *>(dst + c) = x;
        }
    }
}
int main() {
    const size_t sizes[][2] = {{2049, 2048}, {3073, 3072}, {4097, 4096}, {5001, 5008}, {6001, 6000}, {8193, 8192}};
    for (auto &sz : sizes) {
        const size_t rows = sz[0], pitch = sz[1], bytes = rows * pitch * 8;
        double *a, *b, *p;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&p, pitch * 8);
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(p, 0, pitch * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 4; mode++) {
            const bool inplace = mode & 1, nt = mode & 2;
            const int reps = 20;
            for (int i = 0; i < reps + 3; i++) {
                if (i == 3) hipEventRecord(e0);
                double *src = inplace ? a : ((i & 1) ? b : a), *dst = inplace ? a : ((i & 1) ? a : b);
                if (nt) sweep<true><<<256, 1024>>>(src, dst, p, rows, pitch);
                else sweep<false><<<256, 1024>>>(src, dst, p, rows, pitch);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%zux%zu (%.0f MB) %s %s: %.1f us/sweep, %.2f TB/s (read+write)\n", rows, pitch, bytes / 1e6, inplace ? "in-place " : "ping-pong", nt ? "nt" : "  ", 1e3 * ms / reps, 2.0 * bytes / (ms / reps * 1e-3) / 1e12);
        }
        hipFree(a); hipFree(b); hipFree(p);
    }
    return 0;
}
