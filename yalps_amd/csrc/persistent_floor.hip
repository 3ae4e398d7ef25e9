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
    // rows: [2 parities][NB][2 J T] doubles; flags: [2 parities][NB][2] records, then [2 parities][8][2] XCD records, then [NB] XCD ids
    // (zeroed before the launch; epochs count from 1)
    // variant bit 2: the TWO-LEVEL exchange (DESIGN.md 4.2b "XCD-leader"): the first workgroup of every XCD polls the records of
    // its XCD's workgroups, reduces them and raises ONE record per XCD; everybody polls those (at most 8) instead of NB records.
    __shared__ int sh_fail;
    __shared__ unsigned char sh_x[1024 + 8]; // everybody's XCD (1 + id); [NB + x]: XCD x has workgroups
    __shared__ int sh_leader;
    const int tid = threadIdx.x, b = blockIdx.x, NB = gridDim.x, pitch = 2 * J * T;
    unsigned long long *xrec = flags + (size_t)2 * NB * 2;                              // [2][8][2]
    int32_t *xtab = reinterpret_cast<int32_t *>(flags + (size_t)2 * NB * 2 + 2 * 8 * 2); // [NB]
    double2 mine[J];
#pragma unroll
    for (int j = 0; j < J; j++) mine[j] = make_double2((double)(b + j), (double)tid);
    if (tid == 0) sh_fail = 0;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= 7;
    const bool two_level = (variant & 4) != 0;
    if (two_level && tid == 0) { // my XCD, out before my first record
        __hip_atomic_store(xtab + b, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    bool table_ready = false, leader = false;
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
        // (3) flat: everybody polls everybody.  Two-level (from the second round on: the first one also carries the XCD table):
        // the XCD's leader polls its XCD's records and raises the XCD's record, everybody polls the XCD records.
        const bool flat = !two_level || !table_ready;
        if (tid < NB && (flat || (leader && sh_x[tid] == xcc + 1))) {
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
        if (!flat) {
            if (leader) {
                __syncthreads(); // (the reduction of my XCD's records would sit here)
                if (tid == 0)
                    st16_sc1(reinterpret_cast<double *>(xrec + ((size_t)par * 8 + xcc) * 2),
                             make_double2(mine[0].x, __longlong_as_double((long long)(((unsigned long long)(unsigned)epoch << 32) | (unsigned)b))));
            }
            if (tid < 8 && sh_x[NB + tid]) { // (sh_x[NB + x]: XCD x has workgroups)
                unsigned spins = 0;
                unsigned long long spin_t0 = 0;
                for (;;) {
                    const double2 rec = ld16_sc1_one(xrec + ((size_t)par * 8 + tid) * 2);
                    if ((unsigned)((unsigned long long)__double_as_longlong(rec.y) >> 32) == (unsigned)epoch) break;
                    if (spin_expired(spins, spin_t0, err)) {
                        sh_fail = 1;
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (sh_fail) return;
        if (two_level && !table_ready) { // everybody's XCD (published before the first record, which I have seen by now)
            if (tid < NB) sh_x[tid] = (unsigned char)__hip_atomic_load(xtab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 8) sh_x[NB + tid] = 0;
            __syncthreads();
            if (tid == 0) {
                int first = -1;
                for (int w_ = 0; w_ < NB; w_++) {
                    if (sh_x[w_] >= 1 && sh_x[w_] <= 8) sh_x[NB + sh_x[w_] - 1] = 1;
                    if (first < 0 && sh_x[w_] == xcc + 1) first = w_;
                }
                sh_leader = first == b ? 1 : 0;
            }
            __syncthreads();
            leader = sh_leader != 0;
            table_ready = true;
        }
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
