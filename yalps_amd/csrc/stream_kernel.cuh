// stream_kernel.cuh -- persistent IN-PLACE pivot loop for tableaux that do not fit on chip
// Part of libyalps_hip.so; included by persistent_stream.hip inside its unnamed namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// stream_kernel<T lanes, J units per lane per row, CHECK = options.checkCycles>: the whole pivot loop in ONE launch (like
// resident_kernel, same exchange protocol), but the rows stay in HBM / Infinity Cache and are
// updated IN PLACE by the one workgroup that owns them (rows b, b + NB, ...).  In place is safe
// here because no workgroup ever reads another workgroup's rows: the pivot row reaches everybody
// as the copy its owner published with its candidate (rc_rows, ping-pong by epoch), the objective
// row is replicated in every workgroup's registers, and a workgroup reads the pivot-column /
// entering-column entries of its own rows behind its own barriers.
// What that buys over the launch-per-pivot kernels (pivot_kernel / wide_kernel, which ping-pong
// the whole tableau between two buffers):
//   * rows whose pivot-column entry is <= 1e-16 in magnitude (src/simplex.ts:31) are neither read
//     nor written -- 98-99.6 % of the rows of the reference's sparse netlib problems per pivot;
//   * one buffer instead of two halves the footprint: a 134 MB tableau streams out of the 256 MB
//     Infinity Cache (8.3 TB/s measured for a bare in-place sweep against 5.3 ping-pong);
//   * no launch boundary and no partial-reduction prologue per pivot; the elimination of the rows
//     that are not my next candidate overlaps the time the flags take to travel.
// Per pivot and workgroup: poll NB flags -> arg-min -> winner's raw row (sc1 loads) -> my rows'
// pivot-column entries (gather) -> normalised pivot row to LDS (FLUSHED marks zeroed entries) ->
// objective replica, pricing -> my rows' entries of the next entering column AFTER this pivot
// (computed from the gathered scalars, no row is streamed for it) and RHS -> my candidate -> that
// one row eliminated, stored in place and published -> flag -> the other touched rows.
// The launch leaves the tableau in the buffer it found it in.  A hand-off that gives up (never
// expected) leaves it half updated: the host keeps a copy made before the launch and falls back.
// ------------------------------------------------------------------------------------------
template <int T, int J, bool CHECK>
__global__ __launch_bounds__(T) void stream_kernel(Desc d, int parity, int chunk) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_q, sh_c0; // quotient; objective-row entry of the pivot column
    __shared__ int sh_fail, sh_nt, sh_flag, sh_verdict;
    extern __shared__ double sm_dyn[]; // prow[pitch], colv[rpw], lav[rpw], rhsv[rpw], tlist[rpw] (int)

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
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
    double *mat = d.mat[mbuf];
    double *rhs = d.rhs[mbuf];
    int phase = Sin->phase;
    double iter = Sin->iter;
    int64_t pivots = Sin->pivots;
    int64_t hist_len = Sin->hist_len; // checkCycles: pivots recorded in the current phase
    constexpr bool check_cycles = CHECK; // (a template parameter: <1024,4> has no register to spare for it)
    int slot = 0;
    const int rpw = (d.hcap + NB - 1) / NB;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    double *prow = sm_dyn, *colv = prow + pitch, *lav = colv + rpw, *rhsv = lav + rpw;
    int *tlist = reinterpret_cast<int *>(rhsv + rpw);
    const double flushed = __longlong_as_double((long long)FLUSHED);

    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    // ---- my replica of the objective row (registers), my rows' RHS (LDS) ----
    double2 o[J];
#pragma unroll
    for (int j = 0; j < J; j++) o[j] = *reinterpret_cast<const double2 *>(mat + cofs[j]);
    for (int i = tid; i < my_rows; i += T) rhsv[i] = rhs[b + NB * i];
    if (tid == 0) sh_fail = 0;
    __syncthreads();

    int la = 0; // entering column of the NEXT pivot (phase 2), priced on my objective replica
    auto price = [&]() __attribute__((always_inline)) { // src/simplex.ts:71-79
        KI best = {INFINITY, INT_MAX};
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double ov = elem(o[j], k);
                if (c0 + k < n && ov > precision && ki_better(-ov, c0 + k + 1, best.k, best.i)) {
                    best.k = -ov;
                    best.i = c0 + k + 1;
                }
            }
        }
        best = block_argmin<T>(best, sk, si, slot);
        slot ^= 1;
        la = best.i == INT_MAX ? 0 : best.i;
    };
    // my candidate of the given kind (1 = most negative RHS, 2 = min ratio against lav[]); uniform result
    auto candidate = [&](int kind) __attribute__((always_inline)) {
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
            } else if (la > 0) {
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
    unsigned epoch = 0;
    // flag half of the publication (the row data of slot cg has been stored to rc_rows[par] by the caller)
    auto raise_flag = [&](KI cand, int cg, int par) __attribute__((always_inline)) {
        if (tid == 0) st_sc1(d.rc_key[par] + b, rhsv[cg]); // the candidate row's RHS entry
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
        __syncthreads();                                    // ... before ONE lane raises the flag:
        if (tid == 0) // ONE 16-byte record {candidate key, epoch << 32 | row}, one store, polled with one 16-byte load
            st16_sc1(reinterpret_cast<double *>(d.rc_flag[par] + 2 * b),
                     make_double2(cand.k, __longlong_as_double((long long)(((unsigned long long)epoch << 32) | (unsigned)cand.i))));
    };
    // publish a candidate whose row is taken from the tableau as it stands
    auto publish_from_tableau = [&](KI cand) __attribute__((always_inline)) {
        epoch++;
        const int par = epoch & 1, cg = cand.i == INT_MAX ? 0 : cand.i / NB;
        if (my_rows > 0) {
            const double *mr = mat + (size_t)(b + NB * cg) * pitch;
            double *dst = d.rc_rows[par] + (size_t)b * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 < pitch) st16_sc1(dst + c0, *reinterpret_cast<const double2 *>(mr + c0));
            }
        }
        raise_flag(cand, cg, par);
    };
    // entries of my rows in column la, as the rows are now -> lav[]
    auto column_la = [&]() __attribute__((always_inline)) {
        if (la > 0)
            for (int i = tid; i < my_rows; i += T) lav[i] = ld_sc1(mat + (size_t)(b + NB * i) * pitch + la - 1);
        __syncthreads();
    };
    int done = 0, term = RUNNING;
    double term_result = NAN;
    bool stop = false;
    auto check = [&]() __attribute__((always_inline)) { // src/simplex.ts:69,109 and :80
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

    // first round: candidates from the tableau as loaded
    price();
    column_la();
    check();
    if (!stop) publish_from_tableau(candidate(phase));

    while (!stop) {
        // ---------------- gather everyone's candidate -------------------------------------------
        const int par = epoch & 1;
        KI c = {INFINITY, INT_MAX};
        if (tid < NB) {
            unsigned long long f = 0;
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            double2 rec;
            for (;;) {
                rec = ld16_sc1_one(d.rc_flag[par] + 2 * tid);
                f = (unsigned long long)__double_as_longlong(rec.y);
                if ((unsigned)(f >> 32) == epoch) break;
                if (spin_expired(spins, spin_t0, d.rc_err)) {
                    sh_fail = 1;
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            c.i = (int)(unsigned)f;
            c.k = rec.x;
        }
        c = block_argmin<T>(c, sk, si, slot); // (its barrier is the one the polling waves join)
        slot ^= 1;
        if (sh_fail) return; // uniform: written before the barrier above
        if (c.i == INT_MAX) {
            if (phase == 1) { // :120 phase 1 is over: same tableau, now the min-ratio exchange
                phase = 2;
                iter = 0.0;
                hist_len = 0;
                check();
                if (!stop) {
                    column_la();
                    publish_from_tableau(candidate(2));
                }
            } else {
                term = YALPS_UNBOUNDED; // :96
                term_result = (double)la;
                stop = true;
            }
            continue;
        }
        const int row = c.i, owner = row % NB;
        if ((unsigned)row >= (unsigned)h) { // (never expected: a record that names no row of this tableau -- leave with the error
            if (tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // word set instead of indexing with it)
            return;
        }
        const int lslot = owner == b ? row / NB : -1; // my slot of the pivot row, if I own it
        // ---------------- the winner's raw row (sc1 loads only) ----------------------------------
        const double *src = d.rc_rows[par] + (size_t)owner * pitch;
        const double rhs_row = ld_sc1(d.rc_key[par] + owner);
        double2 pv[J];
        ld16_sc1<J>(pv, src, cofs);
        int col = la;
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
            e = block_argmin<T>(e, sk, si, slot);
            slot ^= 1;
            if (e.i == INT_MAX) { // :135
                term = YALPS_INFEASIBLE;
                stop = true;
                continue;
            }
            col = e.i;
        }
        if (check_cycles) { // :98,137 hasCycle before the pivot: workgroup 0 (it maintains the basis) decides for everybody
            int cycled = 0;
            if (b == 0) { // (the basis is updated by my lane 0 behind my own barriers; sc1 loads read it at L2)
                const int leaving = __hip_atomic_load(d.var + w + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int entering = __hip_atomic_load(d.var + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cycled = has_cycle(C, hist_len, leaving, entering, &sh_flag) ? 1 : 0;
                if (tid == 0)
                    __hip_atomic_store(d.rc_verdict + par, ((unsigned long long)epoch << 32) | (unsigned)cycled,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) {
                    unsigned long long v = 0;
                    unsigned spins = 0;
            unsigned long long spin_t0 = 0;
                    for (;;) {
                        v = __hip_atomic_load(d.rc_verdict + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(v >> 32) == epoch) break;
                        if (spin_expired(spins, spin_t0, d.rc_err)) {
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
        // ---------------- pivot (src/simplex.ts:5-39) ---------------------------------------------
        const int colx = col - 1, ucol = colx >> 1, ecol = colx & 1, col_tid = ucol % T, col_j = ucol / T;
        // pivot-column entries of my rows (gather; my own rows, complete since my last barrier), the
        // objective row's entry and the quotient (from the lane that holds that column)
        // (sc1 loads: other waves of this workgroup stored these rows last pivot; read them at L2, not from a vector-L1 line)
        for (int i = tid; i < my_rows; i += T) colv[i] = ld_sc1(mat + (size_t)(b + NB * i) * pitch + colx);
        if (tid == col_tid) {
#pragma unroll
            for (int j = 0; j < J; j++)
                if (j == col_j) {
                    sh_c0 = elem(o[j], ecol);
                    sh_q = elem(pv[j], ecol);
                }
        }
        __syncthreads();
        const double q = sh_q, coef0 = sh_c0, inv_q = 1.0 / q;
        // normalised pivot row -> LDS (:14-25); pv keeps the normalised values (0.0 where flushed)
        unsigned nzmask = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double v = elem(pv[j], k);
                const bool nz = fabs(v) > 1e-16;
                pv[j] = with_elem(pv[j], k, nz ? v / q : 0.0);
                if (nz) nzmask |= 1u << (2 * j + k);
            }
            if (c0 < pitch)
                *reinterpret_cast<double2 *>(prow + c0) = make_double2((nzmask & (1u << (2 * j))) ? pv[j].x : flushed,
                                                                       (nzmask & (1u << (2 * j + 1))) ? pv[j].y : flushed);
        }
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
        for (int i = tid; i < my_rows; i += T) { // RHS entries of my rows (:33 at column 0)
            const double coef = colv[i];
            if (i == lslot)
                rhsv[i] = pn_rhs;
            else if (fabs(coef) > 1e-16 && nz_rhs) {
                const double prod = coef * pn_rhs;
                rhsv[i] = rhsv[i] - prod;
            }
        }
        if (fabs(coef0) > 1e-16) { // my replica of the objective row
#pragma unroll
            for (int j = 0; j < J; j++) {
                const double px = coef0 * pv[j].x, py = coef0 * pv[j].y;
                const double nx = o[j].x - px, ny = o[j].y - py;
                o[j].x = (nzmask & (1u << (2 * j))) ? nx : o[j].x;
                o[j].y = (nzmask & (1u << (2 * j + 1))) ? ny : o[j].y;
                if (tid == col_tid && j == col_j) o[j] = with_elem(o[j], ecol, -coef0 / q); // :36
            }
        }
        iter += 1.0;
        pivots += 1;
        done += 1;
        price(); // la of the next pivot (its barrier also publishes prow / rhsv to the workgroup)
        check();

        // one row of mine, as pivot() leaves it: stored in place, optionally also to `pub` (my publication)
        auto finish_row = [&](int i, double *pub) __attribute__((always_inline)) {
            const double coef = colv[i];
            const bool is_piv = i == lslot;
            const bool act = !is_piv && fabs(coef) > 1e-16; // :31
            double *mr = mat + (size_t)(b + NB * i) * pitch;
            if (!is_piv && !act && pub == nullptr) return;
            const double nq = -coef / q;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 >= pitch) continue;
                double2 v;
                if (is_piv) {
                    v = pv[j];
                    if (tid == col_tid && j == col_j) v = with_elem(v, ecol, inv_q); // :25
                } else {
                    v = *reinterpret_cast<const double2 *>(mr + c0);
                    if (act) {
                        const double px = coef * pv[j].x, py = coef * pv[j].y;
                        const double nx = v.x - px, ny = v.y - py;
                        v.x = (nzmask & (1u << (2 * j))) ? nx : v.x;
                        v.y = (nzmask & (1u << (2 * j + 1))) ? ny : v.y;
                        if (tid == col_tid && j == col_j) v = with_elem(v, ecol, nq); // :36
                    }
                }
                if (is_piv || act) *reinterpret_cast<double2 *>(mr + c0) = v;
                if (pub) st16_sc1(pub + c0, v);
            }
        };
        // the same for a row whose entries are already in registers (streaming loop below)
        auto finish_loaded = [&](int i, double2 (&x)[J]) __attribute__((always_inline)) {
            const double coef = colv[i];
            const bool is_piv = i == lslot;
            double *mr = mat + (size_t)(b + NB * i) * pitch;
            const double nq = -coef / q;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 >= pitch) continue;
                double2 v;
                if (is_piv) {
                    v = pv[j];
                    if (tid == col_tid && j == col_j) v = with_elem(v, ecol, inv_q); // :25
                } else { // (every row of the list but the pivot row has |coef| > 1e-16)
                    v = x[j];
                    const double px = coef * pv[j].x, py = coef * pv[j].y;
                    const double nx = v.x - px, ny = v.y - py;
                    v.x = (nzmask & (1u << (2 * j))) ? nx : v.x;
                    v.y = (nzmask & (1u << (2 * j + 1))) ? ny : v.y;
                    if (tid == col_tid && j == col_j) v = with_elem(v, ecol, nq); // :36
                }
                *reinterpret_cast<double2 *>(mr + c0) = v;
            }
        };
        int cg = -1;
        if (!stop) {
            if (phase == 2) {
                // my rows' entries of column la AFTER this pivot, from the gathered scalars
                const int lax = la - 1;
                const double p = prow[lax]; // normalised pivot-row entry of column la, or FLUSHED
                const bool pnz = (unsigned long long)__double_as_longlong(p) != FLUSHED;
                for (int i = tid; i < my_rows; i += T) {
                    double v = ld_sc1(mat + (size_t)(b + NB * i) * pitch + lax);
                    const double coef = colv[i];
                    if (i == lslot)
                        v = la == col ? inv_q : (pnz ? p : 0.0);
                    else if (fabs(coef) > 1e-16) {
                        if (la == col)
                            v = -coef / q;
                        else if (pnz) {
                            const double prod = coef * p;
                            v = v - prod;
                        }
                    }
                    lav[i] = v;
                }
                __syncthreads();
            }
            const KI cand = candidate(phase);
            cg = cand.i == INT_MAX ? 0 : cand.i / NB;
            epoch++;
            const int pnext = epoch & 1;
            if (my_rows > 0) finish_row(cg, d.rc_rows[pnext] + (size_t)b * pitch);
            raise_flag(cand, cg, pnext);
        }
        // ---------------- the other rows: only the touched ones are streamed ----------------------
        if (tid < 64) { // compact list of my touched rows (wave 0)
            int cnt = 0;
            for (int base = 0; base < my_rows; base += 64) {
                const int i = base + tid;
                const bool t = i < my_rows && i != cg && (i == lslot || fabs(colv[i]) > 1e-16);
                const unsigned long long m = __ballot(t);
                if (t) tlist[cnt + __popcll(m & ((1ull << tid) - 1ull))] = i;
                cnt += __popcll(m);
            }
            if (tid == 0) sh_nt = cnt;
        }
        __syncthreads();
        const int nt = sh_nt;
        if constexpr (J <= 2) { // two rows in flight per lane (the registers allow it up to J = 2)
            for (int k = 0; k < nt; k += 2) {
                const int i0 = tlist[k], i1 = tlist[k + 1 < nt ? k + 1 : k];
                const double *m0 = mat + (size_t)(b + NB * i0) * pitch, *m1 = mat + (size_t)(b + NB * i1) * pitch;
                double2 xa[J], xb[J];
#pragma unroll
                for (int j = 0; j < J; j++) xa[j] = *reinterpret_cast<const double2 *>(m0 + cofs[j]);
#pragma unroll
                for (int j = 0; j < J; j++) xb[j] = *reinterpret_cast<const double2 *>(m1 + cofs[j]);
                finish_loaded(i0, xa);
                if (k + 1 < nt) finish_loaded(i1, xb);
            }
        } else {
            for (int k = 0; k < nt; k++) finish_row(tlist[k], nullptr);
        }
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12 (off the critical path)
            const int leaving = d.var[w + row], entering = d.var[col];
            __hip_atomic_store(d.var + w + row, entering, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.var + col, leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d.pos[leaving] = col;
            d.pos[entering] = w + row;
        }
        __syncthreads(); // my rows are complete (and colv / prow / tlist free) before the next round reads them
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
