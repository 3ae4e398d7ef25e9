// stream2_kernel.cuh -- persistent in-place pivot loop with the row updates DELAYED: up to d.delay_depth pivots per sweep
// Part of libyalps_hip.so; included by persistent_stream2.hip inside its unnamed namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// stream2_kernel<T lanes, J units per lane per row, NT>: stream_kernel's protocol and data placement (rows in HBM /
// Infinity Cache, updated in place by the workgroup that owns them; objective replica in registers; one candidate
// exchange per pivot with the candidate rows published write-through) -- but a pivot's elimination (src/simplex.ts:27-38)
// is not carried out on the rows when the pivot is decided.  It stays PENDING: its normalised pivot row stays in LDS, my
// rows' pivot-column entries stay in LDS, and everything the NEXT decision needs is computed from scalars, as stream_kernel
// already does for its look-ahead:
//   * my rows' right-hand sides (:33 at column 0) and my objective replica are updated at once;
//   * my rows' entries of any one column c "as of now" = the entry in memory, run through the pending pivots' scalar form
//     x - coef * p[c] (with the 1e-16 flush, the -coef/quotient and 1/quotient patches of :14-25, :36): one gather, a few
//     flops per row;
//   * my candidate row for the next exchange is loaded, run through the pending pivots in registers and published -- it
//     alone, not stored.
// When d.delay_depth pivots are pending (2 .. 8: as many normalised pivot rows as fit in LDS; or the loop stops) every
// touched row is streamed ONCE and gets all pending eliminations in registers, oldest first, each with its own separately
// rounded multiply and subtract in the reference's order: bit for bit what that many sweeps leave, at 1/depth of the
// traffic.  The exchange is unchanged: one per pivot.
// Per pivot an element costs 16 / depth bytes of HBM traffic instead of 16: where stream_kernel / sweep_kernel sit on the
// memory roofline (4097 x 4097: 38 us per pivot out of the Infinity Cache, 8193 x 8193: 183 us out of HBM) this one moves
// a half to an eighth; what remains per pivot is the exchange (~10 us).
// No checkCycles (those solves keep stream_kernel).  NT: non-temporal row traffic (tableaux beyond the Infinity Cache).
// ------------------------------------------------------------------------------------------
template <int T, int J, bool NT>
__global__ __launch_bounds__(T) void stream2_kernel(Desc d, int parity, int chunk) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_q, sh_c0; // quotient; objective-row entry of the pivot column
    __shared__ int sh_fail, sh_nt;
    constexpr int MAXD = 4;
    __shared__ int sh_pl[MAXD], sh_pc[MAXD]; // the pending pivots, oldest first: my slot of the pivot row (-1: not mine), pivot column (mat index)
    __shared__ int sh_fast[MAXD][T / 64];       // per wave: nothing of its slice of that pivot row was flushed (:31 select-free path)
    extern __shared__ double sm_dyn[]; // prow[depth][pitch], colv[depth][rpw], nqv[depth][rpw], lav[rpw], rhsv[rpw], tlist[rpw] (int)

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
    const int64_t hist_len = Sin->hist_len;
    int slot = 0;
    const int rpw = (d.hcap + NB - 1) / NB;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    const int depth = d.delay_depth < 1 ? 1 : d.delay_depth > MAXD ? MAXD : d.delay_depth;
    double *prow0 = sm_dyn, *colv0 = prow0 + (size_t)depth * pitch, *nqv0 = colv0 + (size_t)depth * rpw, *lav = nqv0 + (size_t)depth * rpw,
           *rhsv = lav + rpw; // (nqv: what replaces a row's pivot-column entry, :25 / :36 -- one division per row and pivot, by one lane)
    int *tlist = reinterpret_cast<int *>(rhsv + rpw);
    const double flushed = __longlong_as_double((long long)FLUSHED);

    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    unsigned padmask = 0; // columns of mine that do not exist (c0 + k >= n): 0.0 in a pivot row, must not count as "flushed"
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (2 * (tid + j * T) + k >= n) padmask |= 1u << (2 * j + k);
    constexpr unsigned FULL = (1u << (2 * J)) - 1u;
    // ---- my replica of the objective row (registers), my rows' RHS (LDS) ----
    double2 o[J];
#pragma unroll
    for (int j = 0; j < J; j++) o[j] = *reinterpret_cast<const double2 *>(mat + cofs[j]);
    for (int i = tid; i < my_rows; i += T) rhsv[i] = rhs[b + NB * i];
    if (tid == 0) sh_fail = 0;
    __syncthreads();

    // ---- the pending pivots (npend of them, [0] the oldest): scalars, pivot rows and my rows' pivot-column entries in LDS ----
    int npend = 0;
    // entry (my row slot i, mat column c) after ONE pending pivot, given the entry before it (:14-25, :31-36 for one element)
    auto after1 = [&](const double *prowp, const double *colvp, const double *nqvp, int lslotp, int colxp, int i, double v, int c)
                      __attribute__((always_inline)) {
        const double p = prowp[c];
        const bool pnz = (unsigned long long)__double_as_longlong(p) != FLUSHED;
        const double coef = colvp[i];
        if (i == lslotp) return c == colxp ? nqvp[i] : (pnz ? p : 0.0);
        if (fabs(coef) > 1e-16) {
            if (c == colxp) return nqvp[i];
            if (pnz) {
                const double prod = coef * p;
                return v - prod;
            }
        }
        return v;
    };
    // my rows' entries of mat column c as they are NOW (memory + the pending pivots) -> out[]; one barrier
    auto column_now = [&](int c, double *out) __attribute__((always_inline)) {
        for (int i = tid; i < my_rows; i += T) {
            double v = ld_sc1(mat + (size_t)(b + NB * i) * pitch + c);
            for (int p = 0; p < npend; p++) v = after1(prow0 + (size_t)p * pitch, colv0 + p * rpw, nqv0 + p * rpw, sh_pl[p], sh_pc[p], i, v, c);
            out[i] = v;
        }
        __syncthreads();
    };
    // one pending pivot applied to a whole row slice held in registers (the same arithmetic as stream_kernel's finish_row)
    auto apply_row = [&](const double *prowp, bool is_piv, double coef, double patch, int colxp, bool fastp, double2 (&x)[J])
                         __attribute__((always_inline)) {
        const bool act = !is_piv && fabs(coef) > 1e-16; // :31
        if (!is_piv && !act) return;
        if (fastp && !is_piv) { // nothing of this wave's slice was flushed: two fp64 instructions per element, no select
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 >= pitch) continue;
                const double2 pn = *reinterpret_cast<const double2 *>(prowp + c0);
                const double px = coef * pn.x, py = coef * pn.y;
                x[j].x = x[j].x - px;
                x[j].y = x[j].y - py;
                if (c0 == (colxp & ~1)) {
                    if (colxp & 1)
                        x[j].y = patch;
                    else
                        x[j].x = patch;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (c0 >= pitch) continue;
            const double2 pn = *reinterpret_cast<const double2 *>(prowp + c0);
            const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
            const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
            if (is_piv) {
                x[j].x = f0 ? pn.x : 0.0;
                x[j].y = f1 ? pn.y : 0.0;
            } else {
                const double px = coef * pn.x, py = coef * pn.y;
                const double nx = x[j].x - px, ny = x[j].y - py;
                x[j].x = f0 ? nx : x[j].x;
                x[j].y = f1 ? ny : x[j].y;
            }
            if (c0 == (colxp & ~1)) {
                if (colxp & 1)
                    x[j].y = patch;
                else
                    x[j].x = patch;
            }
        }
    };
    // (the pending pivots' wave-uniform scalars are read once per use of apply_pending's caller into registers -- p is a
    // compile-time index there --; per row and pivot two LDS words remain: the row's coefficient and its patch value.  With
    // every scalar re-read and -coef/quotient re-divided per row and pivot, by every wave, a row cost 0.85 us per pending
    // pivot: more than its memory traffic)
    int pcx[MAXD], pls[MAXD];
    bool pfast[MAXD];
    auto pending_scalars = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < MAXD; p++) {
            pcx[p] = p < npend ? sh_pc[p] : 0;
            pls[p] = p < npend ? sh_pl[p] : -1;
            pfast[p] = p < npend ? sh_fast[p][tid >> 6] != 0 : false;
        }
    };
    auto apply_pending = [&](int i, double2 (&x)[J]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < MAXD; p++)
            if (p < npend) // (uniform)
                apply_row(prow0 + (size_t)p * pitch, i == pls[p], colv0[p * rpw + i], nqv0[p * rpw + i], pcx[p], pfast[p], x);
    };
    auto load_row = [&](int i, double2 (&x)[J]) __attribute__((always_inline)) {
        const double *m = mat + (size_t)(b + NB * i) * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) x[j] = ld_row(m + cofs[j], NT);
    };
    auto store_row = [&](int i, const double2 (&x)[J]) __attribute__((always_inline)) {
        double *m = mat + (size_t)(b + NB * i) * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (c0 >= pitch) continue;
            if (NT)
                st_row_nt(m + c0, x[j]);
            else
                *reinterpret_cast<double2 *>(m + c0) = x[j];
        }
    };
    // every touched row streamed once, all pending eliminations in registers; afterwards nothing is pending
    auto flush_pending = [&]() __attribute__((always_inline)) {
        if (npend == 0) return; // (uniform)
        if (tid < 64) { // compact list of my touched rows (wave 0)
            int cnt = 0;
            for (int base = 0; base < my_rows; base += 64) {
                const int i = base + tid;
                bool t = false;
                if (i < my_rows)
                    for (int p = 0; p < npend; p++) t = t || i == sh_pl[p] || fabs(colv0[p * rpw + i]) > 1e-16;
                const unsigned long long m = __ballot(t);
                if (t) tlist[cnt + __popcll(m & ((1ull << tid) - 1ull))] = i;
                cnt += __popcll(m);
            }
            if (tid == 0) sh_nt = cnt;
        }
        __syncthreads();
        const int nt = sh_nt;
        pending_scalars();
        // NBUF row buffers taking turns: NBUF - 1 rows' loads are in flight while one row gets its eliminations and is stored
        // (a wave's rows are a dependent load -> compute -> store chain each: with one row ahead the sweep ran at the
        // latency of a row, 2 us, not at the bandwidth of the memory: 4097 x 4097 spent 1.0 us per row and pivot)
        constexpr int NBUF = J <= 2 ? 4 : 3;
        {
            double2 xb[NBUF][J];
#pragma unroll
            for (int u = 0; u < NBUF - 1; u++)
                if (u < nt) load_row(tlist[u], xb[u]);
            for (int k = 0; k < nt; k += NBUF) {
#pragma unroll
                for (int u = 0; u < NBUF; u++) {
                    if (k + u + NBUF - 1 < nt) load_row(tlist[k + u + NBUF - 1], xb[(u + NBUF - 1) % NBUF]);
                    if (k + u < nt) {
                        const int i0 = tlist[k + u];
                        apply_pending(i0, xb[u]);
                        store_row(i0, xb[u]);
                    }
                }
            }
        }
        npend = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // my rows are complete in memory (and the LDS arrays free) before anything reads them again
    };

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
    // my candidate and its row AS IT IS NOW (memory + pending pivots, in registers; not stored): write-through, drained, flag
    auto publish = [&](KI cand) __attribute__((always_inline)) {
        epoch++;
        const int par = epoch & 1, cg = cand.i == INT_MAX ? 0 : cand.i / NB;
        if (my_rows > 0) {
            double2 x[J];
            const double *m = mat + (size_t)(b + NB * cg) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) x[j] = *reinterpret_cast<const double2 *>(m + cofs[j]);
            pending_scalars();
            apply_pending(cg, x);
            double *dst = d.rc_rows[par] + (size_t)b * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 < pitch) st16_sc1(dst + c0, x[j]);
            }
        }
        if (tid == 0) st_sc1(d.rc_key[par] + b, rhsv[cg]); // the candidate row's RHS entry
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
        __syncthreads();                                    // ... before ONE lane raises the flag:
        if (tid == 0) // ONE 16-byte record {candidate key, epoch << 32 | row}, one store, polled with one 16-byte load
            st16_sc1(reinterpret_cast<double *>(d.rc_flag[par] + 2 * b),
                     make_double2(cand.k, __longlong_as_double((long long)(((unsigned long long)epoch << 32) | (unsigned)cand.i))));
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
    if (la > 0)
        column_now(la - 1, lav);
    else
        __syncthreads();
    check();
    if (!stop) publish(candidate(phase));

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
                check();
                if (!stop) {
                    if (la > 0)
                        column_now(la - 1, lav);
                    else
                        __syncthreads();
                    publish(candidate(2));
                }
            } else {
                term = YALPS_UNBOUNDED; // :96
                term_result = (double)la;
                stop = true;
            }
            continue;
        }
        const int row_in = c.i, row = (unsigned)row_in < (unsigned)h ? row_in : 0, owner = row % NB;
        if (row != row_in && tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (never expected)
        const int lslot = owner == b ? row / NB : -1; // my slot of the pivot row, if I own it
        // ---------------- the winner's row as published: the tableau's row after every earlier pivot ----------------
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
        // ---------------- pivot (src/simplex.ts:5-39): it becomes pending pivot number npend ---------------------------
        const int colx = col - 1, ucol = colx >> 1, ecol = colx & 1, col_tid = ucol % T, col_j = ucol / T;
        double *prowN = prow0 + (size_t)npend * pitch, *colvN = colv0 + npend * rpw, *nqvN = nqv0 + npend * rpw;
        // my rows' pivot-column entries as they are now (gather + the older pending pivots), the objective row's entry and
        // the quotient (from the lane that holds that column)
        if (tid == col_tid) {
#pragma unroll
            for (int j = 0; j < J; j++)
                if (j == col_j) {
                    sh_c0 = elem(o[j], ecol);
                    sh_q = elem(pv[j], ecol);
                }
        }
        if (phase == 2) {
            // (phase 2 enters column la, whose entries of my rows "as of now" are what the look-ahead computed for the ratio
            // test: the same routine on the same pending pivots, or -- after a flush -- the same arithmetic carried out on
            // the rows; no gather, one trip through the fabric less on every pivot's chain)
            for (int i = tid; i < my_rows; i += T) colvN[i] = lav[i];
            __syncthreads(); // (also publishes sh_q / sh_c0)
        } else {
            column_now(colx, colvN); // (its barrier also publishes sh_q / sh_c0)
        }
        const double q = sh_q, coef0 = sh_c0, inv_q = 1.0 / q;
        // normalised pivot row -> LDS (:14-25); pv keeps the normalised values (0.0 where flushed) for the objective replica
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
                *reinterpret_cast<double2 *>(prowN + c0) = make_double2((nzmask & (1u << (2 * j))) ? pv[j].x : flushed,
                                                                        (nzmask & (1u << (2 * j + 1))) ? pv[j].y : flushed);
        }
        // The doubles behind column n of a device row are padding (rows are 128 bytes apart): the pass above has marked them
        // FLUSHED like any other zero; the select-free path of the sweep multiplies every lane's units by this row, so the
        // lane that holds them overwrites its own marks with a finite 0.0 (same lane, same address: in order).
        {
            const int u_first = n >> 1, d_lane = (tid - u_first) & (T - 1); // (T is a power of two)
            if (d_lane < (pitch >> 1) - u_first) {
                const int c0p = 2 * (u_first + d_lane);
                if (c0p >= n) prowN[c0p] = 0.0;
                prowN[c0p + 1] = 0.0;
            }
        }
        {
            const bool fast = __builtin_amdgcn_ballot_w64(((nzmask | padmask) & FULL) != FULL) == 0; // (per wave)
            if ((tid & 63) == 0) sh_fast[npend][tid >> 6] = fast ? 1 : 0;
        }
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
        for (int i = tid; i < my_rows; i += T) { // RHS entries of my rows (:33 at column 0)
            const double coef = colvN[i];
            if (i == lslot)
                rhsv[i] = pn_rhs;
            else if (fabs(coef) > 1e-16 && nz_rhs) {
                const double prod = coef * pn_rhs;
                rhsv[i] = rhsv[i] - prod;
            }
            nqvN[i] = i == lslot ? inv_q : -coef / q; // what replaces the pivot column (:25, :36)
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
        if (tid == 0) { // (published to the workgroup by price()'s barrier, like prow / rhsv)
            sh_pl[npend] = lslot;
            sh_pc[npend] = colx;
        }
        npend += 1;
        iter += 1.0;
        pivots += 1;
        done += 1;
        price(); // la of the next pivot (its barrier also publishes prow / rhsv to the workgroup)
        check();
        if (!stop) {
            if (phase == 2) column_now(la - 1, lav); // my rows' entries of column la after every pivot so far
            publish(candidate(phase));               // (la > 0 here: check() stops phase 2 without an entering column)
        }
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12 (off the critical path)
            const int leaving = d.var[w + row], entering = d.var[col];
            d.var[w + row] = entering;
            d.var[col] = leaving;
            d.pos[leaving] = col;
            d.pos[entering] = w + row;
        }
        // ---------------- the rows: only every depth-th pivot (or on the way out) ----------------------------------------
        if (npend == depth || stop)
            flush_pending();
        else
            __syncthreads(); // (colv / rhsv / lav of this pivot are complete before the next round's lanes read them)
    }
    flush_pending(); // (a pivot decided before a break out of the loop: unbounded / infeasible leave with one pending)

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
