// resident2_kernel.cuh -- persistent kernel, tableau resident in the register files: second generation of the pivot loop
// Part of libyalps_hip.so; included by the persistent_resident2_*.hip translation units inside their unnamed namespaces
// (gfx950 only), after resident_kernel.cuh (whose sc1 load / store helpers and hand-off protocol it shares).
#pragma once

// ------------------------------------------------------------------------------------------
// Same data layout, same exchange (Guideline 16 R1: candidate key + candidate row per workgroup, ping-pong by epoch
// parity, 16-byte flag records), same decisions bit for bit as resident_kernel<T, J, R> -- what changed is the work
// INSIDE a workgroup between two exchanges, which in-kernel stage stamps (profiles/r02_resident_stages_gen1.json)
// showed to be 5.5 of the 6.9 us of a pivot at 2049 x 2049 (the exchange itself: 1.3 us):
//   * phase 2 carries the pivot column of my rows from one pivot to the next: the look-ahead of pivot k computes my rows'
//     entries of the next entering column as they will be AFTER pivot k -- those ARE the pivot-column entries of pivot
//     k+1 (src/simplex.ts:28-36 recomputed by the same two roundings) -- so pivot k+1 starts normalising as soon as the
//     winner's row is in: no gather through LDS, no barrier; the quotient is one more 8-byte load of the winner's row, the
//     objective row's entry of the column is the pricing key;
//   * the look-ahead is spread over lanes: the lane that holds the column only deposits its R raw entries in LDS (one
//     ds_write each), lane g of wave 0 does row g's arithmetic together with its ratio (was: one lane, R rows in turn);
//   * the elimination has a wave-uniform fast path for "no pivot-row entry of this wave was flushed" (src/simplex.ts:18-23:
//     always so on dense tableaux): two fp64 instructions per element instead of nine (the selects on the non-zero mask and
//     on "is this the pivot column" are gone; the pivot column is patched afterwards by the one lane that holds it);
//   * reductions: raw v_min_f64 (no canonicalising v_max in front of every step), DPP row broadcasts for the cross-row
//     step, ballot + readlane for the index, log2(waves) steps in the second level;
//   * five workgroup barriers per pivot instead of eight; the candidate row is stored from the registers it was just
//     computed in.
// Variants: plain rows in registers only (the LDS-row and tagged-granule variants stay on resident_kernel).
// ------------------------------------------------------------------------------------------

// v_min_f64 as the hardware has it (keys are never NaN): the builtin fmin costs a canonicalising v_max_f64 per operand
__device__ __forceinline__ double min_f64_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32_rm(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_rm(double v) {
    const int lo = dpp_i32_rm<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp_i32_rm<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// every lane of a 16-lane row gets the minimum of the row's first 2^STEPS-lane group it belongs to
template <int STEPS>
__device__ __forceinline__ double rowN_min_raw(double v) {
    v = min_f64_raw(v, dpp_f64<DPP_XOR1>(v));
    if constexpr (STEPS >= 2) v = min_f64_raw(v, dpp_f64<DPP_XOR2>(v));
    if constexpr (STEPS >= 3) v = min_f64_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    if constexpr (STEPS >= 4) v = min_f64_raw(v, dpp_f64<DPP_MIRROR>(v));
    return v;
}
template <int STEPS>
__device__ __forceinline__ int rowN_min(int v) {
    v = min(v, dpp_i32<DPP_XOR1>(v));
    if constexpr (STEPS >= 2) v = min(v, dpp_i32<DPP_XOR2>(v));
    if constexpr (STEPS >= 3) v = min(v, dpp_i32<DPP_HALF_MIRROR>(v));
    if constexpr (STEPS >= 4) v = min(v, dpp_i32<DPP_MIRROR>(v));
    return v;
}
// 64-lane (key, index) arg-min, lowest index among equal keys; result uniform over the wave (the same function of the
// same inputs as wave_argmin)
__device__ __forceinline__ KI wave_argmin2(KI v) {
    double m = rowN_min_raw<4>(v.k);
    m = min_f64_raw(m, dpp_f64_rm<DPP_ROW_BCAST15, 0xA>(m)); // rows 1, 3 <- min(rows 0..1), min(rows 2..3)
    m = min_f64_raw(m, dpp_f64_rm<DPP_ROW_BCAST31, 0xC>(m)); // row 3 <- min of all four
    KI r;
    r.k = lane_f64(m, 63);
    const bool hit = v.k == r.k;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
    if (__popcll(mask) == 1) { // (the usual case: one lane holds the minimum)
        r.i = __builtin_amdgcn_readlane(v.i, __ffsll((long long)mask) - 1);
    } else {
        int i = rowN_min<4>(hit ? v.i : INT_MAX);
        i = min(i, dpp_i32_rm<DPP_ROW_BCAST15, 0xA>(i));
        i = min(i, dpp_i32_rm<DPP_ROW_BCAST31, 0xC>(i));
        r.i = __builtin_amdgcn_readlane(i, 63);
    }
    return r;
}
// Result broadcast to every lane of the workgroup; one barrier (sk / si: [2][16] LDS scratch, `slot` alternates)
template <int T>
__device__ __forceinline__ KI block_argmin2(KI v, double (*sk)[16], int (*si)[16], int slot) {
    constexpr int NW = T / 64, STEPS = NW == 16 ? 4 : NW == 8 ? 3 : NW == 4 ? 2 : 1;
    static_assert(NW == 2 || NW == 4 || NW == 8 || NW == 16, "waves per workgroup");
    v = wave_argmin2(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sk[slot][wv] = v.k;
        si[slot][wv] = v.i;
    }
    __syncthreads();
    const double k = sk[slot][lane & (NW - 1)]; // every group of NW lanes holds all NW wave results
    const int i = si[slot][lane & (NW - 1)];
    // (scalar registers: what follows from the result -- row, column, slot, lane tests -- then compiles to scalar branches
    // instead of v_cmp + exec-mask sequences, which were most of a pivot's instruction stream)
    const double km = rowN_min_raw<STEPS>(k);
    KI r;
    r.k = uniform_f64(km);
    r.i = __builtin_amdgcn_readfirstlane(rowN_min<STEPS>(k == km ? i : INT_MAX));
    return r;
}

template <int T, int J, int R>
__global__ __launch_bounds__(T) void resident2_kernel(Desc d, int parity, int chunk) {
    constexpr int SPLIT = YALPS_SPLIT_NUM * R / 4; // other rows eliminated between the candidate row's stores and its flag
    constexpr unsigned FULL = (1u << (2 * J)) - 1u;
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_cf[2][R + 2]; // pivot-column entries of my rows, by pivot parity ([R], [R + 1]: phase 1's objective entry, quotient)
    __shared__ double sh_nq[R + 2];    // -coef/quotient per row (:36); [R + 1]: 1/quotient (:25)
    __shared__ double sh_raw[R + 2];   // look-ahead: my rows' entries of the next entering column as the rows are now; [R]: the pivot row's
    __shared__ double sh_ck;           // my candidate for the next exchange: key, row, local slot
    __shared__ int sh_ci, sh_cg, sh_fail, sh_flag, sh_verdict, sh_pnz;
    extern __shared__ int sh_perm[]; // workgroup 0: var[perm_len] then pos[perm_len]

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch, w = d.w;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int mbuf = Sin->mbuf;
    const double *matA = d.mat[mbuf];
    const double *rhsA = d.rhs[mbuf];
    int phase = Sin->phase;
    double iter = Sin->iter;
    int64_t pivots = Sin->pivots;
    int64_t hist_len = Sin->hist_len; // checkCycles: pivots recorded in the current phase (src/simplex.ts:67,107)
    const bool check_cycles = C->check_cycles != 0;
    int slot = 0;

    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    // columns of mine that exist (c0 + k < n): a padding column's pivot-row entry is 0.0 = "flushed", which must not
    // keep a wave off the fast path
    unsigned padmask = 0;
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (2 * (tid + j * T) + k >= n) padmask |= 1u << (2 * j + k);
    // ---- load my rows, the objective replica, my rows' RHS (lane g), the basis (workgroup 0) ----
    double2 x[R][J], o[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        o[j] = *reinterpret_cast<const double2 *>(matA + cofs[j]);
        // (the replica is never stored: its padding columns are held at -infinity, which no pricing comparison selects
        // and which the update leaves alone: -inf - coef * 0)
        if (padmask & (1u << (2 * j))) o[j].x = -INFINITY;
        if (padmask & (1u << (2 * j + 1))) o[j].y = -INFINITY;
    }
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int r = b + NB * g;
        const double *mr = matA + (size_t)(r < h ? r : b) * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) x[g][j] = *reinterpret_cast<const double2 *>(mr + cofs[j]);
    }
    const int my_r = b + NB * tid; // lane g = tid < R owns the scalar side of row slot g
    const bool my_live = tid < R && my_r < h;
    double my_rhs = rhsA[my_live ? my_r : 0];
    if (b == 0) {
        for (int i = tid; i < d.perm_len; i += T) {
            sh_perm[i] = d.var[i];
            sh_perm[d.perm_len + i] = d.pos[i];
        }
    }
    if (tid == 0) sh_fail = 0;
    __syncthreads();

    // ---- building blocks of one round ------------------------------------------------------------
    int la = 0;          // entering column of the NEXT pivot (phase 2), priced on my objective replica
    double la_val = 0.0; // ... and the objective row's entry there (the next pivot's coefficient of row 0)
    unsigned epoch = 0;  // exchange round
    int cur = 0;         // sh_cf[cur]: pivot column of my rows for the pivot being applied
    // Dantzig pricing (src/simplex.ts:71-79) on my replica of the objective row -> la, la_val (padding columns: -infinity)
    auto price = [&]() __attribute__((always_inline)) {
        double best = precision; // :72 `value = precision`, strict > below: the first (lowest) column of a lane wins ties
        int bi = INT_MAX;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (o[j].x > best) {
                best = o[j].x;
                bi = c0 + 1;
            }
            if (o[j].y > best) {
                best = o[j].y;
                bi = c0 + 2;
            }
        }
        KI v = {bi == INT_MAX ? INFINITY : -best, bi};
        v = block_argmin2<T>(v, sk, si, slot);
        slot ^= 1;
        la = v.i == INT_MAX ? 0 : v.i; // (scalar registers: block_argmin2)
        la_val = -v.k;
    };
    // Look-ahead, step 1: the lane that holds column la deposits my rows' entries of that column (as the registers hold
    // them now) and, with a pivot pending, the normalised pivot row's entry; one barrier.
    auto deposit_la = [&](const double2 (&pvn)[J], unsigned nzmask, bool pending) __attribute__((always_inline)) {
        const int ula = (la - 1) >> 1, ela = (la - 1) & 1, lt = ula % T, lj = ula / T;
        if (la > 0 && wave == (lt >> 6)) { // (uniform)
#pragma unroll
            for (int j = 0; j < J; j++)
#pragma unroll
                for (int e = 0; e < 2; e++)
                    if (2 * (tid + j * T) + e + 1 == la) { // (one lane of the workgroup, one (j, e): no register is indexed at run time)
#pragma unroll
                        for (int g = 0; g < R; g++) sh_raw[g] = e ? x[g][j].y : x[g][j].x;
                        if (pending) {
                            sh_raw[R] = e ? pvn[j].y : pvn[j].x;
                            sh_pnz = (nzmask >> (2 * j + e)) & 1u;
                        }
                    }
        }
        __syncthreads();
    };
    // lanes 0..R-1: candidate of my row of the given kind (1 = most negative RHS, 2 = min ratio against `value`), reduced
    // over wave 0 and left in sh_ck / sh_ci / sh_cg for every lane; lane g also leaves `value` in sh_cf[which] as row g's
    // pivot-column entry of the pivot this candidate is for
    auto candidate = [&](int kind, double value, int which, bool barrier) __attribute__((always_inline)) {
        if (wave == 0) {
            KI c = {INFINITY, INT_MAX};
            if (my_live && my_r >= 1) {
                if (kind == 1) {
                    if (my_rhs < -precision) {
                        c.k = my_rhs;
                        c.i = my_r;
                    }
                } else if (la > 0 && value > precision) {
                    const double ratio = my_rhs / value;
                    if (ratio < INFINITY) {
                        c.k = (ratio <= precision) ? -INFINITY : ratio;
                        c.i = my_r;
                    }
                }
            }
            if (kind == 2 && tid < R) sh_cf[which][tid] = value;
            c = wave_argmin2(c);
            if (tid == 0) {
                sh_ck = c.k;
                sh_ci = c.i;
                sh_cg = c.i == INT_MAX ? 0 : c.i / NB;
            }
        }
        if (barrier) __syncthreads();
    };
    auto publish_flag = [&]() __attribute__((always_inline)) {
        const int par = epoch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
        __syncthreads();                                  // ... before ONE lane raises the flag:
        if (tid == 0) // ONE 16-byte record {candidate key, epoch << 32 | row}, one store, polled with one 16-byte load
            st16_sc1(reinterpret_cast<double *>(d.rc_flag[par] + 2 * b),
                     make_double2(sh_ck, __longlong_as_double((long long)(((unsigned long long)epoch << 32) | (unsigned)sh_ci))));
    };
    // the candidate row (register slot cg) and its RHS entry, write-through, for round epoch + 1
    auto publish_stores = [&](int cg) __attribute__((always_inline)) {
        epoch++;
        double *dst = d.rc_rows[epoch & 1] + (size_t)b * pitch;
#pragma unroll
        for (int g = 0; g < R; g++) {
            if (g == cg) { // (uniform; the slot index stays a compile-time constant: the rows are registers)
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 < pitch) st16_sc1(dst + c0, x[g][j]);
                }
            }
        }
        if (tid == cg) st_sc1(d.rc_key[epoch & 1] + b, my_rhs); // the candidate row's RHS entry (lane cg)
    };
    int done = 0, term = RUNNING;
    double term_result = NAN;
    bool stop = false;
#ifdef YALPS_STAMPS
    unsigned long long st_acc[20] = {}, st_last = 0, st_t0 = 0, st_r0 = 0;
#endif
    // loop bound, optimality: checked before every exchange (src/simplex.ts:69,109 and :80)
    auto check = [&]() __attribute__((always_inline)) {
        if (done == chunk) {
            stop = true;
        } else if (!(iter < max_pivots)) {
            term = YALPS_CYCLED;
            stop = true;
        } else if (phase == 2 && la == 0) {
            term = YALPS_OPTIMAL;
            stop = true;
        }
    };
    // a round without a pending pivot (launch start, phase switch): candidates from the rows as they are
    const double2 no_row[J] = {};
    auto open_round = [&]() __attribute__((always_inline)) {
        double value = 0.0;
        if (phase == 2) {
            deposit_la(no_row, 0u, false);
            if (tid < R) value = sh_raw[tid];
        }
        candidate(phase, value, cur, true); // (no pivot in flight: the next one reads sh_cf[cur])
        publish_stores(__builtin_amdgcn_readfirstlane(sh_cg));
        publish_flag();
    };

    // first round: candidates from the tableau as loaded
    price();
    check();
    if (!stop) open_round();
#ifdef YALPS_STAMPS
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0)::"memory");
    st_last = st_t0;
#endif
    // (single back edge, single exit: every `stop` is a flag, so the rows stay in one set of registers)
    while (!stop) {
        // ---------------- gather everyone's candidate -------------------------------------------
        const int par = epoch & 1;
        KI c = {INFINITY, INT_MAX};
        if (tid < NB) {
            // key and tag are one 16-byte record, written by one store and read by one load
            unsigned long long f = 0;
            unsigned spins = 0;
            [[maybe_unused]] unsigned long long spin_t0 = 0;
            double2 rec;
            for (;;) {
                rec = ld16_sc1_one(d.rc_flag[par] + 2 * tid);
                f = (unsigned long long)__double_as_longlong(rec.y);
                if ((unsigned)(f >> 32) == epoch) break;
                if (RESIDENT_SPIN_EXPIRED(spins, spin_t0, d.rc_err)) {
                    sh_fail = 1;
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            c.i = (int)(unsigned)f;
            c.k = rec.x;
#ifdef YALPS_AB_POLL_GUARD
            // (never expected: a record that names no row of this tableau -- leave through the failure exit, with the error
            // word set, instead of indexing with it)
            if (c.i != INT_MAX && (unsigned)c.i >= (unsigned)h) {
                sh_fail = 1;
                __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
        }
        YSTAMP(0); // wait for everybody's flag (waves 0 .. NB/64 - 1; the others go straight to the barrier)
        c = block_argmin2<T>(c, sk, si, slot); // (its barrier is the one the polling waves join)
        slot ^= 1;
        YSTAMP(1);
        if (sh_fail) return; // uniform: written before the barrier above
        if (c.i == INT_MAX) {
            if (phase == 1) { // :120 phase 1 is over: same tableau, now the min-ratio exchange
                phase = 2;
                iter = 0.0;
                hist_len = 0;
                check();
                if (!stop) open_round();
            } else {
                term = YALPS_UNBOUNDED; // :96
                term_result = (double)la;
                stop = true;
            }
            YSTAMP(14);
            continue;
        }
#if defined(YALPS_AB_RETURN_GUARD) || defined(YALPS_AB_NOGUARD) || defined(YALPS_AB_POLL_GUARD)
        const int row = c.i, owner = row % NB;
#else
        // (never expected: a record that names no row of this tableau.  No exit of its own -- a `return` here, or a test in
        // the poll loop, cost the tall variants 4-11 % per pivot through register allocation alone, same-box A/B --: the
        // index is clamped, so that nothing is addressed with it, and the error word makes the host discard the launch)
        const int row_in = c.i, row = (unsigned)row_in < (unsigned)h ? row_in : 0, owner = row % NB;
        if (row != row_in && tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#ifdef YALPS_AB_RETURN_GUARD
        if ((unsigned)row >= (unsigned)h) { // (never expected: a record that names no row of this tableau -- leave with the error
            if (tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // word set instead of indexing with it)
            return;
        }
#endif
        // ---------------- the winner's raw row (sc1 loads only) ----------------------------------
        // (phase 2: the quotient M[row, la] rides along as one more 8-byte load of the same row)
        const double *src = d.rc_rows[par] + (size_t)owner * pitch;
        double2 pv[J];
        const double rhs_row = ld_sc1(d.rc_key[par] + owner);
        double q = ld_sc1(src + (phase == 2 ? la - 1 : 0)); // (two loads hipcc counts itself, in flight with the row's)
        ld16_sc1<J>(pv, src, cofs);
        YSTAMP(2); // the winner's row
        int col = la;
        double coef0 = la_val;
        if (phase == 1) { // :123-134
            KI e = {INFINITY, INT_MAX};
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const double coefficient = elem(pv[j], k);
                    if (c0 + k < n && coefficient < -precision) {
                        const double ratio = -elem(o[j], k) / coefficient;
                        if (ratio > -INFINITY && ki_better(-ratio, c0 + k + 1, e.k, e.i)) {
                            e.k = -ratio;
                            e.i = c0 + k + 1;
                        }
                    }
                }
            }
            e = block_argmin2<T>(e, sk, si, slot);
            slot ^= 1;
            if (e.i == INT_MAX) { // :135
                term = YALPS_INFEASIBLE;
                stop = true;
                continue;
            }
            col = e.i;
            // pivot-column entries of my rows, of the objective row, the quotient: from the lane that holds the column
            const int ucol1 = (col - 1) >> 1, ecol1 = (col - 1) & 1;
            if (tid == ucol1 % T) {
#pragma unroll
                for (int j = 0; j < J; j++)
                    if (j == ucol1 / T) {
#pragma unroll
                        for (int g = 0; g < R; g++) sh_cf[cur][g] = elem(x[g][j], ecol1);
                        sh_cf[cur][R] = elem(o[j], ecol1);
                        sh_cf[cur][R + 1] = elem(pv[j], ecol1);
                    }
            }
            __syncthreads();
            coef0 = sh_cf[cur][R];
            q = sh_cf[cur][R + 1];
        }
        if (check_cycles) { // :98,137 hasCycle before the pivot: workgroup 0 (it holds the basis) decides for everybody
            int cycled = 0;
            if (b == 0) {
                const int leaving = sh_perm[w + row], entering = sh_perm[col]; // var[] = sh_perm[0 .. perm_len)
                cycled = has_cycle(C, hist_len, leaving, entering, &sh_flag) ? 1 : 0;
                if (tid == 0)
                    __hip_atomic_store(d.rc_verdict + par, ((unsigned long long)epoch << 32) | (unsigned)cycled,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) {
                    unsigned long long v = 0;
                    unsigned spins = 0;
                    [[maybe_unused]] unsigned long long spin_t0 = 0;
                    for (;;) {
                        v = __hip_atomic_load(d.rc_verdict + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(v >> 32) == epoch) break;
                        if (RESIDENT_SPIN_EXPIRED(spins, spin_t0, d.rc_err)) {
                            sh_fail = 1;
                            __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    sh_verdict = (int)(unsigned)v;
                }
                __syncthreads();
                if (sh_fail) return;
                cycled = sh_verdict;
            }
            hist_len += 1;
            if (cycled) { // ["cycled", NaN]: the tableau stays as it was before this pivot
                term = YALPS_CYCLED;
                stop = true;
                continue;
            }
        }
        YSTAMP(3); // phase 1: entering column + column gather; checkCycles: verdict
        // ---------------- pivot (src/simplex.ts:5-39) on my registers ----------------------------
        const int ucol = (col - 1) >> 1, ecol = (col - 1) & 1, col_tid = ucol % T, col_j = ucol / T;
        const int col_wave = col_tid >> 6;
        double cf[R]; // uniform: pivot-column entry of each of my rows (phase 2: left there by the last look-ahead)
#pragma unroll
        for (int g = 0; g < R; g++) // (scalar registers where the VGPR budget of 2 waves per SIMD is short)
            cf[g] = (T >= 512 && J * R >= 18) ? uniform_f64(sh_cf[cur][g]) : sh_cf[cur][g];
        const double my_coef = tid < R ? sh_cf[cur][tid] : 0.0;
        // rows of mine this pivot changes (:31), one bit per slot, in scalar registers
        const int lane = tid & 63;
        const double lane_cf = sh_cf[cur][lane < R ? lane : 0];
        const unsigned long long touched = __builtin_amdgcn_ballot_w64(lane < R && b + NB * lane < h && fabs(lane_cf) > 1e-16);
        const unsigned long long live = __builtin_amdgcn_ballot_w64(lane < R && b + NB * lane < h);
        const bool all_touched = touched == live;
        // :14-24 normalise; which of my columns were flushed
        unsigned nzmask = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const bool nzx = fabs(pv[j].x) > 1e-16, nzy = fabs(pv[j].y) > 1e-16;
            pv[j].x = nzx ? pv[j].x / q : 0.0;
            pv[j].y = nzy ? pv[j].y / q : 0.0;
            nzmask |= (nzx ? 1u : 0u) << (2 * j) | (nzy ? 1u : 0u) << (2 * j + 1);
        }
        const bool fast = __builtin_amdgcn_ballot_w64(((nzmask | padmask) & FULL) != FULL) == 0; // (uniform over the wave)
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const int lslot = owner == b ? row / NB : -1; // my register slot of the pivot row, if I own it
        // the R + 1 divisions of the pivot column and my rows' RHS entries: lane g of wave 0 for row g
        double my_nq = 0.0;
        if (wave == 0) {
            // ONE division sequence for all three kinds of quotient: lane g < R: -coef_g / q (:36), lane R + 1: 1 / q (:25),
            // lane R + 2: RHS_row / q (:20 at column 0) -- three sequences, one per branch, kept wave 0 behind the others
            const double num = tid < R ? -my_coef : tid == R + 1 ? 1.0 : rhs_row;
            const double quot = num / q;
            my_nq = quot;
            if (tid < R || tid == R + 1) sh_nq[tid] = quot;
            const double rhs_q = lane_f64(quot, R + 2);
            if (my_live) { // RHS entry of my row (:33 at column 0)
                const double pn_rhs = nz_rhs ? rhs_q : 0.0;
                if (tid == lslot)
                    my_rhs = pn_rhs;
                else if (fabs(my_coef) > 1e-16 && nz_rhs) {
                    const double prod = my_coef * pn_rhs;
                    my_rhs = my_rhs - prod;
                }
            }
        }
        if (fabs(coef0) > 1e-16) { // my replica of the objective row
            if (fast) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const double px = coef0 * pv[j].x, py = coef0 * pv[j].y;
                    o[j].x = o[j].x - px;
                    o[j].y = o[j].y - py;
                }
            } else {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const double px = coef0 * pv[j].x, py = coef0 * pv[j].y;
                    const double nx = o[j].x - px, ny = o[j].y - py;
                    o[j].x = (nzmask & (1u << (2 * j))) ? nx : o[j].x;
                    o[j].y = (nzmask & (1u << (2 * j + 1))) ? ny : o[j].y;
                }
            }
            if (wave == col_wave) {
                if (tid == col_tid) { // :36 for row 0 (one more division, in the one lane that needs it)
                    const double nq0 = -coef0 / q;
#pragma unroll
                    for (int j = 0; j < J; j++)
                        if (j == col_j) {
                            if (ecol)
                                o[j].y = nq0;
                            else
                                o[j].x = nq0;
                        }
                }
            }
        }
        YSTAMP(5); // normalise, column divisions, RHS, objective replica
        iter += 1.0;
        pivots += 1;
        done += 1;
        price(); // la of the next pivot, from the updated objective replica (its barrier also publishes sh_nq)
        check();
        YSTAMP(6);
        // my rows in the slots of `set` (bit g), fully, as pivot() leaves them
        auto finish_rows = [&](unsigned long long set) __attribute__((always_inline)) {
            const unsigned long long pivbit = lslot >= 0 ? (1ull << lslot) & set : 0ull; // the pivot row, if it is mine and in the set
            const unsigned long long elim = set & touched & ~pivbit;                      // rows to eliminate (:31)
#pragma unroll
            for (int g = 0; g < R; g++) { // (g must stay a compile-time index: the rows are registers; every test is scalar)
                if ((pivbit >> g) & 1u) {
#pragma unroll
                    for (int j = 0; j < J; j++) x[g][j] = pv[j];
                } else if ((elim >> g) & 1u) {
                    if (fast) {
#pragma unroll
                        for (int j = 0; j < J; j++) {
                            const double px = cf[g] * pv[j].x, py = cf[g] * pv[j].y;
                            x[g][j].x = x[g][j].x - px;
                            x[g][j].y = x[g][j].y - py;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < J; j++) {
                            const double px = cf[g] * pv[j].x, py = cf[g] * pv[j].y;
                            const double nx = x[g][j].x - px, ny = x[g][j].y - py;
                            x[g][j].x = (nzmask & (1u << (2 * j))) ? nx : x[g][j].x;
                            x[g][j].y = (nzmask & (1u << (2 * j + 1))) ? ny : x[g][j].y;
                        }
                    }
                }
            }
            if (wave == col_wave && (elim | pivbit) != 0) { // the pivot column itself (:25, :36): patched by the one lane that holds it
                if (tid == col_tid) {
#pragma unroll
                    for (int g = 0; g < R; g++) {
                        if (!(((elim | pivbit) >> g) & 1u)) continue;
                        const double v = sh_nq[((pivbit >> g) & 1u) ? R + 1 : g];
#pragma unroll
                        for (int j = 0; j < J; j++)
                            if (j == col_j) {
                                if (ecol)
                                    x[g][j].y = v;
                                else
                                    x[g][j].x = v;
                            }
                    }
                }
            }
        };
        constexpr unsigned long long ONE = 1, ALL = (ONE << R) - 1, LOW = (ONE << SPLIT) - 1;
        if (!stop) {
            double value = 0.0;
            if (phase == 2) {
                // my rows' entries of column la AFTER this pivot: the lane that holds the column deposits the raw
                // entries, lane g of wave 0 applies this pivot to row g's
                deposit_la(pv, nzmask, true);
                YSTAMP(7);
                if (tid < R) {
                    value = sh_raw[tid];
                    if (tid == lslot)
                        value = la == col ? sh_nq[R + 1] : sh_raw[R];
                    else if (my_live && fabs(my_coef) > 1e-16) {
                        if (la == col)
                            value = my_nq;
                        else if (sh_pnz) {
                            const double prod = my_coef * sh_raw[R];
                            value = value - prod;
                        }
                    }
                }
            }
            // Dense case -- nothing of this wave's pivot-row slice was flushed and every live row of mine is touched (:31):
            // the candidate row is formed in registers of its own and published; behind the flag ALL rows, the candidate
            // row included, are eliminated as straight-line code (two fp64 instructions per element, no branch per row:
            // the per-row scalar branches of finish_rows cost more in register copies and spill reloads than the
            // arithmetic).  Same-box A/B at 2049 x 2049: 6.55 -> 5.85 us per pivot.  Same barriers as the general path.
            if (fast && all_touched) {
                candidate(phase, value, cur ^ 1, true);
                const int cg = __builtin_amdgcn_readfirstlane(sh_cg);
                // the candidate row after this pivot, in registers of its own (x[cg] is updated with the others below: the
                // same two roundings on the same inputs give the same bits)
                double2 cand[J];
                double ccf = 0.0;
#pragma unroll
                for (int g = 0; g < R; g++)
                    if (g == cg) {
                        ccf = cf[g];
#pragma unroll
                        for (int j = 0; j < J; j++) cand[j] = x[g][j];
                    }
                if (cg == lslot) {
#pragma unroll
                    for (int j = 0; j < J; j++) cand[j] = pv[j];
                } else {
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const double px = ccf * pv[j].x, py = ccf * pv[j].y;
                        cand[j].x = cand[j].x - px;
                        cand[j].y = cand[j].y - py;
                    }
                }
                if (wave == col_wave) {
                    if (tid == col_tid) {
                        const double v = sh_nq[cg == lslot ? R + 1 : cg];
#pragma unroll
                        for (int j = 0; j < J; j++)
                            if (j == col_j) {
                                if (ecol)
                                    cand[j].y = v;
                                else
                                    cand[j].x = v;
                            }
                    }
                }
                epoch++;
                {
                    double *dst = d.rc_rows[epoch & 1] + (size_t)b * pitch;
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const int c0 = 2 * (tid + j * T);
                        if (c0 < pitch) st16_sc1(dst + c0, cand[j]);
                    }
                    if (tid == cg) st_sc1(d.rc_key[epoch & 1] + b, my_rhs);
                }
                publish_flag();
#pragma unroll
                for (int g = 0; g < R; g++)
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const double px = cf[g] * pv[j].x, py = cf[g] * pv[j].y;
                        x[g][j].x = x[g][j].x - px;
                        x[g][j].y = x[g][j].y - py;
                    }
                if (lslot >= 0) {
#pragma unroll
                    for (int g = 0; g < R; g++)
                        if (g == lslot) {
#pragma unroll
                            for (int j = 0; j < J; j++) x[g][j] = pv[j];
                        }
                }
                if (wave == col_wave) {
                    if (tid == col_tid) {
#pragma unroll
                        for (int g = 0; g < R; g++) {
                            if (b + NB * g >= h) continue;
                            const double v = sh_nq[g == lslot ? R + 1 : g];
#pragma unroll
                            for (int j = 0; j < J; j++)
                                if (j == col_j) {
                                    if (ecol)
                                        x[g][j].y = v;
                                    else
                                        x[g][j].x = v;
                                }
                        }
                    }
                }
            } else {
                candidate(phase, value, cur ^ 1, true);
                YSTAMP(8);
                const int cg = __builtin_amdgcn_readfirstlane(sh_cg);
                finish_rows(ONE << cg);
                YSTAMP(9);
                publish_stores(cg);
                YSTAMP(10);
                if constexpr (SPLIT > 0) finish_rows(LOW & ~(ONE << cg)); // (while the stores drain)
                YSTAMP(11);
                publish_flag();
                YSTAMP(12);
                finish_rows(ALL & ~LOW & ~(ONE << cg)); // (while the flags travel)
            }
        } else {
            finish_rows(ALL);
        }
        cur ^= 1;
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12, in LDS (off the critical path)
            int *var = sh_perm, *pos = sh_perm + d.perm_len;
            const int leaving = var[w + row], entering = var[col];
            var[w + row] = entering;
            var[col] = leaving;
            pos[leaving] = col;
            pos[entering] = w + row;
        }
        // (no barrier here: sh_nq / sh_raw / sh_cf[next] are next written behind the gather's barrier, which every wave
        // reaches only after it has finished reading them)
        YSTAMP(13); // the other rows
    }
#ifdef YALPS_STAMPS
    if (tid == 0 && d.dbg) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        unsigned long long *out = d.dbg + (size_t)b * STAMP_WORDS;
#pragma unroll
        for (int k = 0; k < 20; k++) out[k] += st_acc[k];
        out[20] += (unsigned long long)done;
        out[21] += t1 - st_t0;
        out[22] += r1 - st_r0;
    }
#endif

    // ---------------- leave: tableau to the other buffer, state, basis ---------------------------
    double *matB = d.mat[mbuf ^ 1];
    double *rhsB = d.rhs[mbuf ^ 1];
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int r = b + NB * g;
        if (r < h) {
            double *mr = matB + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 < pitch) *reinterpret_cast<double2 *>(mr + c0) = x[g][j];
            }
        }
    }
    if (my_live) rhsB[my_r] = my_rhs;
    if (b == 0) {
        // (the last pivot's basis swap was one lane's LDS writes at the very end of the loop body, with no barrier behind
        // them: a wave that got here first copied the old entries -- seen once in 380 GPU tests, on a 60-pivot solve)
        __syncthreads();
        for (int i = tid; i < d.perm_len; i += T) {
            d.var[i] = sh_perm[i];
            d.pos[i] = sh_perm[d.perm_len + i];
        }
        if (tid == 0) {
            if (term == YALPS_OPTIMAL) term_result = round_to_precision(my_rhs, precision); // lane 0 = row 0
            Sout->status = term;
            Sout->phase = phase;
            Sout->bootstrap = 1; // the streaming kernel would have to re-scan
            Sout->la = 0;
            Sout->pbuf = 0;
            Sout->mbuf = mbuf ^ 1;
            Sout->pause = 0;
            Sout->dec_valid = 0;
            Sout->dec_row = 0;
            Sout->dec_col = 0;
            Sout->swap_valid = 0;
            Sout->swap_row = 0;
            Sout->swap_col = 0;
            Sout->pad_ = 0;
            Sout->hist_len = hist_len;
            Sout->iter = iter;
            Sout->result = term_result;
            Sout->pivots = pivots;
        }
    }
}
