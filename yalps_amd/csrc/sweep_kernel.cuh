// sweep_kernel.cuh -- persistent IN-PLACE pivot loop for tableaux that stream from HBM (rows of 8194 .. 16385 columns)
// Part of libyalps_hip.so; included by persistent_sweep.hip inside its unnamed namespace (gfx950 only), after
// resident_kernel.cuh (sc1 load / store helpers).
#pragma once

// ------------------------------------------------------------------------------------------
// sweep_kernel<T lanes, J 16-byte units per lane and row, CHECK = options.checkCycles>: the whole two-phase loop of
// src/simplex.ts:66-142 in ONE launch for tableaux whose pivot is bound by HBM (a pivot moves 16*h*w bytes: 4.3 GB at
// 16385 x 16385), built around the sweep shape that streams fastest on this chip (tools/micro/sweep_patterns.hip,
// profiles/r02_sweep_patterns.txt: normalised pivot row in REGISTERS -- a lane always meets the same columns --, two
// rows in flight per lane, non-temporal loads and stores once the tableau is beyond the Infinity Cache: 6.18 TB/s at
// 16385 x 16384 against 5.25 for the plain one-row-at-a-time loop).
//
// Rows b, b + NB, ... belong to workgroup b (one per CU) and are updated in place by it alone; nobody keeps a replica of
// anything, the objective row is simply row 0 of workgroup 0.  What a pivot costs besides the sweep is three small
// exchanges (Guideline 16 R1: sc1 stores, drain, barrier, one 16-byte record; sc1 polls and loads), microseconds against
// a sweep of 100 .. 800:
//   E1  every workgroup's candidate {key, row} (min ratio in column la / most negative RHS)  -> arg-min -> pivot row
//   E2  that row's owner publishes the RAW row (+ its RHS entry); everybody loads its slice and normalises it (:14-24)
//   E3  workgroup 0 eliminates row 0 first, prices it from its registers (:71-79) and publishes the NEXT entering column
//       (and, in phase 1, the entering column of THIS pivot, :123-134; with checkCycles the hasCycle verdict, :44-63)
//       while everybody else is still sweeping
// Only rows whose pivot-column entry exceeds 1e-16 (:31) are read or written (compact list, as in stream_kernel).
// A hand-off that gives up leaves the tableau half updated: the host keeps a copy made before the launch (yalps_hip.hip).
// ------------------------------------------------------------------------------------------
struct SweepSync {                   // in the zeroed control block (yalps_hip.hip), 16-byte records {double, epoch << 32 | int}
    unsigned long long row_flag[2];  // E2: epoch of the row published in rc_rows[par]
    unsigned long long rec_la[2][2]; // E3: {objective entry at la, epoch << 32 | la}: the entering column after pivot `epoch - 1`
    unsigned long long rec_col[2][2]; // phase 1 / checkCycles: {cycled ? 1.0 : 0.0, epoch << 32 | col} for the pivot of this epoch
};                                    // (d.sw_recs: [2 parities][nb] x {phase-1 half, phase-2 half}, 32 bytes per workgroup, E1)

// 16-byte row accesses through a buffer descriptor of ONE row (base = the row, a scalar; size = the row: units past the
// pitch are dropped by the range check, no branch) + a 32-bit lane offset: one address register per unit for all rows in
// flight, where flat addressing held a 64-bit pair per unit and row (and spilled them).  aux 2 = non-temporal.
// Unit j of a lane sits at byte 16 * lane + 16 * T * j of its row: one lane-offset register, the unit's part added per access.
// (NOT passed as the instruction's scalar offset: hipcc leaves out the wait state between a 16-byte buffer store and a
// following VALU write of its data registers when the store has an SGPR offset -- the ISA manual's exemption -- and on
// gfx950 the low dword of 16 lanes of a store then went to memory overwritten: every ~25th row of a sweep, bit-exact
// tests caught it.  With a literal 0 there the hazard wait is emitted.)
// AUX: 0 plain, 2 non-temporal, 16 sc1 (the hand-offs' write-through stores and L1-bypassing loads, Guideline 16 R1).
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
constexpr int AUX_PLAIN = 0, AUX_NT = 2, AUX_SC1 = 16;
template <int AUX>
__device__ __forceinline__ double2 row_ld16(__amdgpu_buffer_rsrc_t rs, int lane_off, int unit_off) {
    union {
        v4u32 u;
        double2 v;
    } c;
    c.u = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, unit_off, AUX);
    return c.v;
}
template <int AUX>
__device__ __forceinline__ void row_st16(__amdgpu_buffer_rsrc_t rs, int lane_off, int unit_off, double2 v) {
    union {
        v4u32 u;
        double2 v;
    } c;
    c.v = v;
    __builtin_amdgcn_raw_buffer_store_b128(c.u, rs, lane_off, unit_off, AUX);
}

template <int T, int J, bool CHECK, bool NT>
__global__ __launch_bounds__(T) void sweep_kernel(Desc d, int parity, int chunk) {
    // rows in flight per lane: two where the registers allow it (8 units: 6.07 against 5.79 TB/s measured with one); 16-unit
    // rows in pairs need 128 registers for the rows alone and spill beside the rest of the loop
    constexpr int D = J <= 8 ? 2 : 1;
    static_assert(J <= 16, "one non-zero flag per column of a lane in 32 bits");
    constexpr unsigned FULL = J == 16 ? 0xFFFFFFFFu : (1u << (2 * (J & 15))) - 1u;
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ int sh_fail, sh_nt, sh_flag;
    extern __shared__ __attribute__((aligned(16))) double sw_dyn[]; // (16-byte aligned base: its rows are read and written 16 bytes per lane, Guideline 17)
    // prow[pitch], colv[rpw], lav[rpw], rhsv[rpw], nqv[rpw], tlist[rpw] (int)

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    // (the state is read with vector loads -- the kernel also writes that array --: moved to scalar registers here, a
    // 1024-lane workgroup has 128 vector registers per lane and the sweep needs them for rows in flight)
    const int h = __builtin_amdgcn_readfirstlane(C->height), n = d.n, pitch = d.pitch, w = d.w;
    const double precision = uniform_f64(C->precision), max_pivots = uniform_f64(C->max_pivots);
    const int mbuf = __builtin_amdgcn_readfirstlane(Sin->mbuf);
    double *mat = d.mat[mbuf];
    double *rhs = d.rhs[mbuf];
    SweepSync *sync = reinterpret_cast<SweepSync *>(d.sw_sync);
    int phase = __builtin_amdgcn_readfirstlane(Sin->phase);
    double iter = uniform_f64(Sin->iter);
    int64_t pivots = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(Sin->pivots >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)Sin->pivots));
    int64_t hist_len = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(Sin->hist_len >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)Sin->hist_len));
    int slot = 0;
    const int rpw = (d.hcap + NB - 1) / NB;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    // The normalised pivot row lives in LDS, every lane's 2 J doubles at the columns it also holds of a row (16-byte
    // accesses, conflict free; read and written by that lane only: no barrier): measured equal to registers (6.15 against
    // 6.18 TB/s) and it frees 4 J registers per lane, without which two rows in flight do not fit the 128 of a 1024-lane group.
    // (prow spans all 2 T J doubles the lanes can address, whatever the pitch: every lane writes its units unguarded)
    double *prow = sw_dyn, *colv = prow + 2 * T * J, *lav = colv + rpw, *rhsv = lav + rpw, *nqv = rhsv + rpw;
    const double flushed = __longlong_as_double((long long)FLUSHED);
    int *tlist = reinterpret_cast<int *>(nqv + rpw);

    unsigned padmask = 0; // columns of mine that do not exist (c0 + k >= n): 0.0 in the pivot row, must not count as "flushed"
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (c0 + k >= n) padmask |= 1u << (2 * j + k);
    }
    const int lane_off = 16 * tid, row_bytes = pitch * 8;
    // (the descriptor is built from scalar registers -- readfirstlane of a pointer every lane holds --: one the compiler
    // cannot prove uniform gets every buffer access wrapped in a waterfall loop, CDNA guide T20)
    auto rsrc_of = [&](const double *row_ptr) __attribute__((always_inline)) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(row_ptr);
        const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(u), 0, row_bytes, 0x00020000);
    };
    for (int i = tid; i < my_rows; i += T) rhsv[i] = rhs[b + NB * i];
    if (tid == 0) sh_fail = 0;
    __syncthreads();

    // one 16-byte record {x, epoch << 32 | v}: store (one lane, after the workgroup's stores have drained) / poll
    auto put_rec = [&](unsigned long long *rec, double x, unsigned epoch_, int v) __attribute__((always_inline)) {
        st16_sc1(reinterpret_cast<double *>(rec), make_double2(x, __longlong_as_double((long long)(((unsigned long long)epoch_ << 32) | (unsigned)v))));
    };
    // lane 0 waits for the record of `epoch_`, the workgroup joins a barrier; returns false when the wait gave up
    __shared__ double sh_rx;
    __shared__ int sh_rv;
    auto get_rec = [&](const unsigned long long *rec, unsigned epoch_, double &x, int &v) __attribute__((always_inline)) {
        if (tid == 0) {
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            for (;;) {
                const double2 r = ld16_sc1_one(rec);
                const unsigned long long f = (unsigned long long)__double_as_longlong(r.y);
                if ((unsigned)(f >> 32) == epoch_) {
                    sh_rx = r.x;
                    sh_rv = (int)(unsigned)f;
                    break;
                }
                if (spin_expired(spins, spin_t0, d.rc_err)) {
                    sh_fail = 1;
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        x = sh_rx;
        v = __builtin_amdgcn_readfirstlane(sh_rv);
        return sh_fail == 0;
    };
    // Dantzig pricing (:71-79) of an objective-row slice held in registers (padding columns excluded), units [u0, u0 + JN)
    // at a time (a 16-unit row is priced in two halves: the whole row beside the sweep's buffers does not fit the registers):
    // price_part accumulates a lane's best (value, which of its 2 J columns -- a compile-time constant per comparison: the
    // column numbers themselves, 2 J loop invariants per lane, were hoisted out of the pivot loop into registers the sweep
    // needs), price_finish reduces over the workgroup -> (la, entry)
    constexpr int JP = J > 8 ? J / 2 : J;
    auto price_part = [&](auto &o, int u0, double &best, int &bj) __attribute__((always_inline)) {
        constexpr int JN = (int)(sizeof(o) / sizeof(o[0]));
#pragma unroll
        for (int j = 0; j < JN; j++) {
            if (!(padmask & (1u << (2 * (u0 + j)))) && o[j].x > best) {
                best = o[j].x;
                bj = 2 * (u0 + j);
            }
            if (!(padmask & (1u << (2 * (u0 + j) + 1))) && o[j].y > best) {
                best = o[j].y;
                bj = 2 * (u0 + j) + 1;
            }
        }
    };
    auto price_finish = [&](double best, int bj, int &la_out, double &val_out) __attribute__((always_inline)) {
        const int bi = bj < 0 ? INT_MAX : 2 * (tid + (bj >> 1) * T) + (bj & 1) + 1;
        KI v = {bi == INT_MAX ? INFINITY : -best, bi};
        v = block_argmin<T>(v, sk, si, slot);
        slot ^= 1;
        la_out = __builtin_amdgcn_readfirstlane(v.i == INT_MAX ? 0 : v.i);
        val_out = uniform_f64(-v.k);
    };
    // my candidate of the given kind (1 = most negative RHS, 2 = min ratio against lav[]); uniform result
    auto candidate = [&](int kind, int la_) __attribute__((always_inline)) {
        KI c = {INFINITY, INT_MAX};
        for (int i = tid; i < my_rows; i += T) {
            const int r = b + NB * i;
            if (r < 1) continue;
            const double my_rhs = rhsv[i];
            if (kind == 1) {
                if (my_rhs < -precision && ki_better(my_rhs, r, c.k, c.i)) {
                    c.k = my_rhs;
                    c.i = r;
                }
            } else if (la_ > 0) {
                const double value = lav[i];
                if (value > precision) {
                    const double ratio = my_rhs / value;
                    if (ratio < INFINITY) {
                        const double key = (ratio <= precision) ? -INFINITY : ratio;
                        if (ki_better(key, r, c.k, c.i)) {
                            c.k = key;
                            c.i = r;
                        }
                    }
                }
            }
        }
        c = block_argmin<T>(c, sk, si, slot);
        slot ^= 1;
        return c;
    };

    unsigned epoch = 0; // pivot number within this launch, from 1: tags every record of that pivot
    int done = 0, term = RUNNING;
    double term_result = NAN;
    bool stop = false;
    int la = 0;
    double la_val = 0.0;

    // the entering column of the first pivot of this launch: workgroup 0 prices row 0 as loaded
    if (b == 0) {
        const __amdgpu_buffer_rsrc_t r0 = rsrc_of(mat);
        double best = precision;
        int bj = -1;
#pragma unroll
        for (int u0 = 0; u0 < J; u0 += JP) {
            double2 o[JP];
#pragma unroll
            for (int j = 0; j < JP; j++) o[j] = row_ld16<AUX_PLAIN>(r0, lane_off + 16 * T * (u0 + j), 0);
            price_part(o, u0, best, bj);
        }
        price_finish(best, bj, la, la_val);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) put_rec(sync->rec_la[1], la_val, 1u, la);
    }

    while (!stop) {
        epoch++;
        const int par = epoch & 1;
        // ---------------- E3: the entering column priced after the previous pivot -------------------------------
        if (b != 0) {
            if (!get_rec(sync->rec_la[par], epoch, la_val, la)) return;
        }
        if (done == chunk) { // src/simplex.ts:69,109 and :80
            stop = true;
        } else if (!(iter < max_pivots)) {
            term = YALPS_CYCLED;
            stop = true;
        } else if (phase == 2 && la == 0) {
            term = YALPS_OPTIMAL;
            stop = true;
        }
        if (stop) break;
        // ---------------- E1: candidates -> pivot row ---------------------------------------------------------------
        // One exchange carries BOTH of a workgroup's candidates -- most negative RHS (phase 1, :111-119) and min ratio in
        // column la (phase 2, :83-95) -- as two 16-byte halves of one record, each with its own tag: when phase 1 finds no
        // row (:120) the phase-2 decision is already in every workgroup's registers, no second round.
        int row = 0;
        {
            if (la > 0) {
                for (int i = tid; i < my_rows; i += T) lav[i] = ld_sc1(mat + (size_t)(b + NB * i) * pitch + la - 1);
                __syncthreads();
            }
            KI mine1 = {INFINITY, INT_MAX};
            if (phase == 1) mine1 = candidate(1, la);
            const KI mine2 = candidate(2, la);
            unsigned long long *recs = d.sw_recs + (size_t)par * NB * 4;
            if (tid == 0) {
                put_rec(recs + 4 * b, mine1.k, epoch, mine1.i);
                put_rec(recs + 4 * b + 2, mine2.k, epoch, mine2.i);
            }
            KI c1 = {INFINITY, INT_MAX}, c2 = {INFINITY, INT_MAX};
            if (tid < NB) {
                unsigned spins = 0;
                unsigned long long spin_t0 = 0;
                for (;;) {
                    const double2 r1 = ld16_sc1_one(recs + 4 * tid), r2 = ld16_sc1_one(recs + 4 * tid + 2);
                    const unsigned long long f1 = (unsigned long long)__double_as_longlong(r1.y), f2 = (unsigned long long)__double_as_longlong(r2.y);
                    if ((unsigned)(f1 >> 32) == epoch && (unsigned)(f2 >> 32) == epoch) {
                        c1.k = r1.x;
                        c1.i = (int)(unsigned)f1;
                        c2.k = r2.x;
                        c2.i = (int)(unsigned)f2;
                        break;
                    }
                    if (spin_expired(spins, spin_t0, d.rc_err)) {
                        sh_fail = 1;
                        __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (phase == 1) {
                c1 = block_argmin<T>(c1, sk, si, slot);
                slot ^= 1;
            }
            c2 = block_argmin<T>(c2, sk, si, slot);
            slot ^= 1;
            if (sh_fail) return;
            if (phase == 1 && c1.i == INT_MAX) { // :120 phase 1 is over: same tableau, same entering column, the min-ratio decision
                phase = 2;
                iter = 0.0;
                hist_len = 0;
                if (!(iter < max_pivots)) {
                    term = YALPS_CYCLED;
                    stop = true;
                } else if (la == 0) {
                    term = YALPS_OPTIMAL;
                    stop = true;
                }
            }
            if (!stop) {
                row = __builtin_amdgcn_readfirstlane(phase == 1 ? c1.i : c2.i); // (scalar: row bases become scalar addresses)
                if (row == INT_MAX) { // phase 2 without a row: :96
                    term = YALPS_UNBOUNDED;
                    term_result = (double)la;
                    stop = true;
                }
            }
        }
        if (stop) break;
        if ((unsigned)row >= (unsigned)h) { // (never expected)
            if (tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const int owner = row % NB, lslot = owner == b ? row / NB : -1;
        // ---------------- E2: the owner publishes the raw pivot row ------------------------------------------------
        double *rowbuf = d.rc_rows[par]; // (slot 0 of the candidate-row area: one row)
        if (lslot >= 0) {
            const __amdgpu_buffer_rsrc_t rsrc_row = rsrc_of(mat + (size_t)row * pitch), rsrc_pub = rsrc_of(rowbuf);
            double2 raw[J]; // (my own row, complete since my last barrier: all loads in flight, then published write-through)
#pragma unroll
            for (int j = 0; j < J; j++) raw[j] = row_ld16<AUX_SC1>(rsrc_row, lane_off + 16 * T * j, 0);
#pragma unroll
            for (int j = 0; j < J; j++) row_st16<AUX_SC1>(rsrc_pub, lane_off + 16 * T * j, 0, raw[j]);
            if (tid == 0) st_sc1(d.rc_key[par], rhsv[lslot]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&sync->row_flag[par], (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (tid == 0) {
                unsigned spins = 0;
                unsigned long long spin_t0 = 0;
                while (__hip_atomic_load(&sync->row_flag[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)epoch) {
                    if (spin_expired(spins, spin_t0, d.rc_err)) {
                        sh_fail = 1;
                        __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            if (sh_fail) return;
        }
        const double rhs_row = ld_sc1(d.rc_key[par]);
        double2 pv[J];
        {
            const __amdgpu_buffer_rsrc_t rsrc_pub = rsrc_of(rowbuf);
#pragma unroll
            for (int j = 0; j < J; j++) pv[j] = row_ld16<AUX_SC1>(rsrc_pub, lane_off + 16 * T * j, 0);
        }
        // ---------------- phase 1: entering column (:123-134); checkCycles: hasCycle (:44-63) -- workgroup 0 decides -----
        int col = la;
        if (phase == 1 || CHECK) {
            if (b == 0) {
                if (phase == 1) {
                    KI e = {INFINITY, INT_MAX};
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const int c0 = 2 * (tid + j * T);
                        const double2 o = row_ld16<AUX_SC1>(rsrc_of(mat), lane_off + 16 * T * j, 0); // row 0 is mine: complete since my last barrier
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const double coefficient = elem(pv[j], k);
                            if (c0 + k < n && coefficient < -precision) {
                                const double ratio = -elem(o, k) / coefficient;
                                if (ratio > -INFINITY && ki_better(-ratio, c0 + k + 1, e.k, e.i)) {
                                    e.k = -ratio;
                                    e.i = c0 + k + 1;
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    e = block_argmin<T>(e, sk, si, slot);
                    slot ^= 1;
                    col = __builtin_amdgcn_readfirstlane(e.i == INT_MAX ? 0 : e.i);
                }
                int cycled = 0;
                if (CHECK && col > 0) {
                    const int leaving = __hip_atomic_load(d.var + w + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int entering = __hip_atomic_load(d.var + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cycled = has_cycle(C, hist_len, leaving, entering, &sh_flag) ? 1 : 0;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) put_rec(sync->rec_col[par], cycled ? 1.0 : 0.0, epoch, col);
                if (cycled) col = -1;
            } else {
                double cyc;
                if (!get_rec(sync->rec_col[par], epoch, cyc, col)) return;
                if (cyc != 0.0) col = -1;
            }
            if (col == 0) { // :135
                term = YALPS_INFEASIBLE;
                stop = true;
                break;
            }
            if (CHECK) hist_len += 1;
            if (col < 0) { // ["cycled", NaN]: the tableau stays as it was before this pivot
                term = YALPS_CYCLED;
                stop = true;
                break;
            }
        }
        // ---------------- pivot (src/simplex.ts:5-39) ---------------------------------------------------------------------
        const int colx = col - 1;
        const double q = ld_sc1(rowbuf + colx);
        for (int i = tid; i < my_rows; i += T) colv[i] = ld_sc1(mat + (size_t)(b + NB * i) * pitch + colx); // my rows: complete since my last barrier
        unsigned nzmask = 0;
#pragma unroll
        for (int j = 0; j < J; j++) { // :14-24; FLUSHED marks the entries pivot() zeroed (they are not in nonZeroColumns)
            // (straight-line per unit: each comparison's lane mask lives for a few instructions; held across all units
            // -- 4 J scalar registers -- they were most of this kernel's scalar spills)
            const double qx = pv[j].x / q, qy = pv[j].y / q;
            const bool nzx = fabs(pv[j].x) > 1e-16, nzy = fabs(pv[j].y) > 1e-16;
            nzmask |= (nzx ? 1u : 0u) << (2 * j) | (nzy ? 1u : 0u) << (2 * j + 1);
            *reinterpret_cast<double2 *>(prow + 2 * (tid + j * T)) = make_double2(nzx ? qx : flushed, nzy ? qy : flushed);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The doubles behind column n of a device row are padding (rows are 128 bytes apart): the pass above has marked them
        // FLUSHED like any other zero; the select-free path of the sweep multiplies every lane's units by this row, so the
        // lane that holds them overwrites its own marks with a finite 0.0 (same lane, same address: in order).
        {
            const int u_first = n >> 1, d_lane = (tid - u_first) & (T - 1); // (T is a power of two)
            if (d_lane < (pitch >> 1) - u_first) {
                const int c0p = 2 * (u_first + d_lane);
                if (c0p >= n) prow[c0p] = 0.0;
                prow[c0p + 1] = 0.0;
            }
        }
        const bool fast = __builtin_amdgcn_ballot_w64(((nzmask | padmask) & FULL) != FULL) == 0; // no entry of this wave was flushed
        const double inv_q = 1.0 / q; // :25
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
        __syncthreads(); // colv complete
        for (int i = tid; i < my_rows; i += T) { // RHS entries of my rows (:33 at column 0); what replaces the pivot column (:25, :36)
            const double coef = colv[i];
            if (i == lslot)
                rhsv[i] = pn_rhs;
            else if (fabs(coef) > 1e-16 && nz_rhs) {
                const double prod = coef * pn_rhs;
                rhsv[i] = rhsv[i] - prod;
            }
            nqv[i] = i == lslot ? inv_q : -coef / q;
        }
        if (tid < 64) { // compact list of my touched rows (wave 0), ascending: workgroup 0 meets row 0 first
            int cnt = 0;
            for (int base = 0; base < my_rows; base += 64) {
                const int i = base + tid;
                const bool t = i < my_rows && (i == lslot || fabs(colv[i]) > 1e-16);
                const unsigned long long m = __ballot(t);
                if (t) tlist[cnt + __popcll(m & ((1ull << tid) - 1ull))] = i;
                cnt += __popcll(m);
            }
            if (tid == 0) sh_nt = cnt;
        }
        __syncthreads();
        const int ntouch = __builtin_amdgcn_readfirstlane(sh_nt);
        iter += 1.0;
        pivots += 1;
        done += 1;
        // units [u0, u0 + JN) of one row of mine, as pivot() leaves it, from its loaded slice; the new slice stays in x
        // (u0 is a constant at every call site: the offsets fold)
        auto finish_u = [&](int i, int u0, auto &x) __attribute__((always_inline)) {
            constexpr int JN = (int)(sizeof(x) / sizeof(x[0]));
            const double coef = uniform_f64(colv[i]);
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * i) * pitch);
            if (i == lslot) {
#pragma unroll
                for (int j = 0; j < JN; j++) {
                    const double2 pn = *reinterpret_cast<const double2 *>(prow + 2 * (tid + (u0 + j) * T));
                    x[j].x = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED ? pn.x : 0.0;
                    x[j].y = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED ? pn.y : 0.0;
                }
            } else if (fast) {
#pragma unroll
                for (int j = 0; j < JN; j++) {
                    const double2 pn = *reinterpret_cast<const double2 *>(prow + 2 * (tid + (u0 + j) * T));
                    const double px = coef * pn.x, py = coef * pn.y;
                    x[j].x = x[j].x - px;
                    x[j].y = x[j].y - py;
                }
            } else {
#pragma unroll
                for (int j = 0; j < JN; j++) {
                    const double2 pn = *reinterpret_cast<const double2 *>(prow + 2 * (tid + (u0 + j) * T));
                    const double px = coef * pn.x, py = coef * pn.y;
                    const double nx = x[j].x - px, ny = x[j].y - py;
                    x[j].x = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED ? nx : x[j].x;
                    x[j].y = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED ? ny : x[j].y;
                    __builtin_amdgcn_sched_barrier(0); // (unit by unit: interleaved, the differences waiting for their selects spilled)
                }
            }
            const double patch = uniform_f64(nqv[i]); // the pivot column itself (:25, :36)
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const int c0 = 2 * (tid + (u0 + j) * T);
                if (c0 == (colx & ~1)) {
                    if (colx & 1)
                        x[j].y = patch;
                    else
                        x[j].x = patch;
                }
                row_st16<NT ? AUX_NT : AUX_PLAIN>(rs, lane_off + 16 * T * (u0 + j), 0, x[j]);
            }
        };
        auto load_u = [&](int i, int u0, auto &x) __attribute__((always_inline)) {
            constexpr int JN = (int)(sizeof(x) / sizeof(x[0]));
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * i) * pitch);
#pragma unroll
            for (int j = 0; j < JN; j++) x[j] = row_ld16<NT ? AUX_NT : AUX_PLAIN>(rs, lane_off + 16 * T * (u0 + j), 0);
        };
        auto finish = [&](int i, double2 (&x)[J]) __attribute__((always_inline)) { finish_u(i, 0, x); };
        auto load = [&](int i, double2 (&x)[J]) __attribute__((always_inline)) { load_u(i, 0, x); };
        int k0 = 0;
        if (b == 0) {
            // row 0 first: priced from the registers it was just computed in, the next entering column leaves at once
            const bool touched0 = ntouch > 0 && __builtin_amdgcn_readfirstlane(tlist[0]) == 0; // (uniform)
            double best = precision;
            int bj = -1;
#pragma unroll
            for (int u0 = 0; u0 < J; u0 += JP) {
                double2 o[JP];
                if (touched0) {
                    load_u(0, u0, o);
                    finish_u(0, u0, o);
                } else { // (the objective row's pivot-column entry was within 1e-16: row 0 is unchanged)
#pragma unroll
                    for (int j = 0; j < JP; j++) o[j] = row_ld16<AUX_SC1>(rsrc_of(mat), lane_off + 16 * T * (u0 + j), 0);
                }
                price_part(o, u0, best, bj);
            }
            if (touched0) k0 = 1;
            int nla;
            double nval;
            price_finish(best, bj, nla, nval);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) put_rec(sync->rec_la[(epoch + 1) & 1], nval, epoch + 1, nla);
            la = nla;
            la_val = nval;
        }
        // ---------------- the sweep: D rows in flight per lane -----------------------------------------------------------
        if constexpr (D == 2) {
            // two row buffers taking turns: the next row's loads are in flight while this one is eliminated and stored
            // (slot numbers in scalar registers: a row's base is then a scalar address)
            if (k0 < ntouch) {
                double2 xa[J], xb[J];
                load(__builtin_amdgcn_readfirstlane(tlist[k0]), xa);
                for (int k = k0; k < ntouch; k += 2) {
                    const int i0 = __builtin_amdgcn_readfirstlane(tlist[k]);
                    const int i1 = __builtin_amdgcn_readfirstlane(tlist[k + 1 < ntouch ? k + 1 : k]);
                    const int i2 = __builtin_amdgcn_readfirstlane(tlist[k + 2 < ntouch ? k + 2 : k]);
                    if (k + 1 < ntouch) load(i1, xb);
                    finish(i0, xa);
                    if (k + 2 < ntouch) load(i2, xa);
                    if (k + 1 < ntouch) finish(i1, xb);
                }
            }
        } else {
            // 16-unit rows: two HALF-row buffers taking turns the same way (two whole rows do not fit the registers)
            constexpr int JH = J / 2;
            if (k0 < ntouch) {
                double2 xa[JH], xb[JH];
                load_u(__builtin_amdgcn_readfirstlane(tlist[k0]), 0, xa);
                for (int k = k0; k < ntouch; k++) {
                    const int i0 = __builtin_amdgcn_readfirstlane(tlist[k]);
                    const int i1 = __builtin_amdgcn_readfirstlane(tlist[k + 1 < ntouch ? k + 1 : k]);
                    load_u(i0, JH, xb);
                    finish_u(i0, 0, xa);
                    if (k + 1 < ntouch) load_u(i1, 0, xa);
                    finish_u(i0, JH, xb);
                }
            }
        }
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12
            const int leaving = d.var[w + row], entering = d.var[col];
            __hip_atomic_store(d.var + w + row, entering, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.var + col, leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d.pos[leaving] = col;
            d.pos[entering] = w + row;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my rows are in memory ...
        __syncthreads();                                  // ... before any lane of mine gathers from them again
    }

    // ---------------- leave: RHS column, state (the rows are where they were) --------------------
    for (int i = tid; i < my_rows; i += T) rhs[b + NB * i] = rhsv[i];
    if (b == 0 && tid == 0) {
        if (term == YALPS_OPTIMAL) term_result = round_to_precision(rhsv[0], precision);
        Sout->status = term;
        Sout->phase = phase;
        Sout->bootstrap = 1; // the launch-per-pivot kernels would have to re-scan
        Sout->la = 0;
        Sout->pbuf = 0;
        Sout->mbuf = mbuf;
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
