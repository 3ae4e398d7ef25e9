// persistent_floor.hip -- exchange_floor_kernel: the bare exchange of the register-resident kernels, measured (bench.py's
// `roofline.onchip_floor`): what one pivot of resident_kernel / resident2_kernel cannot go below on THIS chip, whatever its
// arithmetic costs.  One workgroup per CU, `epochs` rounds of exactly the hand-off of resident_kernel.cuh (Guideline 16 R1):
//   (1) PAYLOAD: every workgroup stores a row of 2 J T doubles write-through (sc1), every storing wave drains, barrier;
//   (2) ONE lane stores the workgroup's 16-byte record {key, epoch << 32 | id} (sc1);
//   (3) lanes 0 .. NB-1 poll one record each (16-byte sc1 loads + s_sleep) until it carries the epoch; barrier;
//   (4) FETCH: every lane loads its J units of the "winner's" row (sc1) and the next round's payload depends on them.
// No tableau, no arithmetic beyond keeping the dependencies: (3) alone is the flag round trip, (2)-(4) what a consumer
// waits for, (1)-(4) the whole exchange of a pivot.  Same bounded spins as the product kernels (a foreign kernel on the
// chip makes the launch give up, reported through *err).
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>

#include "../../include/yalps_hip.h"
#include "persistent_tables.h"

namespace {
#include "common.cuh"

#include "resident_kernel.cuh" // (the sc1 load / store helpers, spin_expired)

template <int T, int J>
__global__ __launch_bounds__(T) void exchange_floor_kernel(double *rows, unsigned long long *flags, int32_t *err, double *sink, int epochs,
                                                           int variant) {
    // rows: [2 parities][NB][2 J T] doubles; flags: [2 parities][NB][2] (zeroed before the launch; epochs count from 1)
    __shared__ int sh_fail;
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x, pitch = 2 * J * T;
    double2 mine[J];
#pragma unroll
    for (int j = 0; j < J; j++) mine[j] = make_double2((double)(b + j), (double)tid);
    if (tid == 0) sh_fail = 0;
    __syncthreads();
    for (int epoch = 1; epoch <= epochs; epoch++) {
        const int par = epoch & 1;
        double *myrow = rows + ((size_t)par * NB + b) * pitch;
        if (variant & 1) { // (1) payload, write-through, drained before the flag
#pragma unroll
            for (int j = 0; j < J; j++) st16_sc1(myrow + 2 * (tid + j * T), mine[j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid == 0) // (2)
            st16_sc1(reinterpret_cast<double *>(flags + ((size_t)par * NB + b) * 2),
                     make_double2(mine[0].x, __longlong_as_double((long long)(((unsigned long long)(unsigned)epoch << 32) | (unsigned)b))));
        if (tid < NB) { // (3)
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            for (;;) {
                const double2 rec = ld16_sc1_one(flags + ((size_t)par * NB + tid) * 2);
                if ((unsigned)((unsigned long long)__double_as_longlong(rec.y) >> 32) == (unsigned)epoch) break;
                if (spin_expired(spins, spin_t0, err)) {
                    sh_fail = 1;
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (sh_fail) return;
        if (variant & 2) { // (4) the winner's row: a different workgroup every round, the same for everybody
            const int winner = (int)(((unsigned)epoch * 97u + 13u) % (unsigned)NB);
            const double *src = rows + ((size_t)par * NB + winner) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const double2 v = ld16_sc1_one(src + 2 * (tid + j * T));
                mine[j].x = mine[j].x * 0.5 + v.x; // (the next payload depends on what was fetched)
                mine[j].y = mine[j].y * 0.5 + v.y;
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < J; j++) acc += mine[j].x + mine[j].y;
    sink[(size_t)b * T + tid] = acc;
}
} // namespace

const void *yalps_exchange_floor_fn(int lanes, int units) {
    if (lanes == 512 && units == 2) return reinterpret_cast<const void *>(&exchange_floor_kernel<512, 2>);
    if (lanes == 512 && units == 3) return reinterpret_cast<const void *>(&exchange_floor_kernel<512, 3>);
    if (lanes == 256 && units == 1) return reinterpret_cast<const void *>(&exchange_floor_kernel<256, 1>);
    if (lanes == 256 && units == 2) return reinterpret_cast<const void *>(&exchange_floor_kernel<256, 2>);
    return nullptr;
}
