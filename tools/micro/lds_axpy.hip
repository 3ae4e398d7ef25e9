// Micro-benchmark: the inner loop of panel_flush (yalps_amd/csrc/panel_flush.cuh) with nothing around it -- what does ONE pending
// pivot cost a wave that holds D row segments of 8 units (1024 columns) in registers, the pending row's units coming from LDS?
//   x[d][u] = x[d][u] - coef[d] * p[u]   (product and difference rounded separately), u = 0..7 (16-byte units), d = 0..D-1
// 256 workgroups x T lanes, K pending rows of 512 units in LDS (filled once), REP passes over the K pending rows, no global
// memory traffic inside the timed loop.  Prints cycles per (pending pivot, wave) and the share of the fp64 issue peak.
//   MODE 0: as panel_flush (LDS reads half a turn ahead)   MODE 1: the pending row's units from registers (no LDS in the loop)
//   MODE 2: LDS reads only (no arithmetic on them beyond one add, to keep them alive)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_axpy lds_axpy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <int T, int D, int MODE>
__global__ __launch_bounds__(T) void axpy_kernel(double *__restrict__ out, const double *__restrict__ pend, const double *__restrict__ coef, int K, int rep) {
    constexpr int PU = 512, U = 8, UH = 4;
    extern __shared__ __attribute__((aligned(16))) double lds[]; // [K][2 PU] panel, then [K][64] coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *cf = lds + (size_t)K * 2 * PU;
    for (int i = tid; i < K * PU; i += T) *reinterpret_cast<double2 *>(lds + 2 * (size_t)i) = *reinterpret_cast<const double2 *>(pend + 2 * (size_t)i);
    for (int i = tid; i < K * 64; i += T) cf[i] = coef[i];
    __syncthreads();
    double2 x[D][U];
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int u = 0; u < U; u++) x[d][u] = double2{1.0 + tid + d, 2.0 + u};
    const double *pan = lds + 2 * lane;
    for (int r = 0; r < rep; r++) {
        double2 pa[UH], pb[UH];
        double ca[D], cb[D];
        auto rd_units = [&](int p, int ub, double2 (&pn)[UH]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < UH; u++) pn[u] = *reinterpret_cast<const double2 *>(pan + (size_t)p * 2 * PU + 2 * (ub + u) * 64);
        };
        auto rd_cf = [&](int p, double (&c)[D]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; d++) c[d] = cf[p * 64 + ((wave * D + d) & 63)];
        };
        auto work = [&](const double (&c)[D], int ub, const double2 (&pn)[UH]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; d++)
#pragma unroll
                for (int u = 0; u < UH; u++) {
                    if (MODE == 2) {
                        if (d == 0) x[d][ub + u].x = x[d][ub + u].x + pn[u].x;
                    } else {
                        const double px = c[d] * pn[u].x, py = c[d] * pn[u].y;
                        x[d][ub + u].x = x[d][ub + u].x - px;
                        x[d][ub + u].y = x[d][ub + u].y - py;
                    }
                }
        };
        rd_cf(0, ca);
        rd_units(0, 0, pa);
        if (MODE == 1) rd_units(0, UH, pb);
#pragma unroll 1
        for (int p = 0; p < K; p++) {
            const int pnx = p + 1 < K ? p + 1 : p;
            if (MODE != 1) rd_units(p, UH, pb);
            __builtin_amdgcn_sched_barrier(0);
            work(ca, 0, pa);
            __builtin_amdgcn_sched_barrier(0);
            rd_cf(pnx, cb);
            if (MODE != 1) rd_units(pnx, 0, pa);
            __builtin_amdgcn_sched_barrier(0);
            work(ca, UH, pb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < D; d++) ca[d] = cb[d];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int u = 0; u < U; u++) acc += x[d][u].x + x[d][u].y;
    out[(size_t)blockIdx.x * T + tid] = acc;
}

template <int T, int D, int MODE>
static void run(int K, int rep, double *out, const double *pend, const double *coef) {
    const size_t lds = sizeof(double) * ((size_t)K * 1024 + (size_t)K * 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&axpy_kernel<T, D, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    axpy_kernel<T, D, MODE><<<256, T, lds>>>(out, pend, coef, K, 2);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        axpy_kernel<T, D, MODE><<<256, T, lds>>>(out, pend, coef, K, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double us = best * 1e3, per = us / ((double)rep * K);              // us per pending pivot (all waves of a workgroup in parallel)
    const double ops = (MODE == 2 ? 0.0 : 4.0 * 8 * D * 64 * (T / 64) * 256.0); // fp64 lane-operations per pending pivot, whole chip
    std::printf("T=%4d D=%d K=%2d %-28s %8.1f us  %.3f us per pending pivot = %5.0f cycles at 2.4 GHz   fp64 lane-ops/s %.1f T (peak 39.3 T at one per lane and clock)\n", T, D, K,
                MODE == 0 ? "LDS reads + arithmetic" : MODE == 1 ? "arithmetic only (registers)" : "LDS reads only", us, per, per * 2400.0, ops / per * 1e-6);
}

int main() {
    const int KMAX = 16;
    double *out, *pend, *coef;
    hipMalloc(&out, sizeof(double) * 256 * 1024);
    hipMalloc(&pend, sizeof(double) * KMAX * 1024);
    hipMalloc(&coef, sizeof(double) * KMAX * 64);
    std::vector<double> h(KMAX * 1024, 1e-3), c(KMAX * 64, 0.5);
    hipMemcpy(pend, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(coef, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice);
    const int rep = 400;
    run<512, 2, 0>(16, rep, out, pend, coef);
    run<512, 2, 1>(16, rep, out, pend, coef);
    run<512, 2, 2>(16, rep, out, pend, coef);
    run<512, 4, 0>(16, rep, out, pend, coef);
    run<512, 4, 1>(16, rep, out, pend, coef);
    run<256, 2, 0>(16, rep, out, pend, coef);
    run<256, 2, 1>(16, rep, out, pend, coef);
    run<256, 4, 0>(16, rep, out, pend, coef);
    run<256, 4, 1>(16, rep, out, pend, coef);
    run<256, 4, 2>(16, rep, out, pend, coef);
    return 0;
}
