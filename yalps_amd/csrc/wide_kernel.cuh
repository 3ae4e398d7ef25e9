// wide_kernel.cuh -- streaming pivot for tableaux too wide / tall for register batches (pivot row in LDS)
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// wide_kernel<T lanes, J units per lane per row>: the streaming pivot for tableaux whose rows are
// too wide (or whose workgroups own too many rows) to keep a batch of rows plus the pivot row and
// the objective row in registers (n > 4096 columns: 16385-wide row shards; or > 16 rows per
// workgroup: 4097^2).  Same launch protocol, state, partials and modes (FUSED / APPLY / SHARD)
// as pivot_kernel; the differences are the data flow:
//   * the normalised pivot row lives in LDS (8 B per column, <= 131 KB; FLUSHED marks the
//     entries pivot() zeroed) and is read back 16 B per lane and row (LDS rate >> HBM rate);
//   * the objective row is streamed (twice at most) instead of held;
//   * my rows are streamed ONE at a time, double-buffered: the loads of row i+1 are in flight
//     while row i is eliminated and stored -- register use is independent of the row count;
//   * the scalar side of every row (RHS entry, -coef/quotient) is computed uniformly by all
//     lanes; the rows' entries of the next entering column and their new RHS are parked in LDS
//     and turned into this workgroup's partial after the last row.
// wide_kernel<T, J, true, NT> -- MODE_SHARD only -- sweeps the rows IN PLACE (one tableau buffer, no second 2 GB of DRAM
// pages in play; rows with |coef| <= 1e-16, src/simplex.ts:31, are loaded for the look-ahead but never stored).  That is
// safe in this mode because no workgroup reads another workgroup's rows: the pivot row arrives in the all-gathered slot,
// my rows' pivot-column entries and RHS are gathered into LDS behind a barrier BEFORE the first row is stored (the waves
// of a workgroup stream their column slices at their own pace: a wave that loaded them with the row, as the ping-pong form
// does, found the entry already replaced by -coef/quotient, :36, by the wave that owns that column), and the
// one row everybody reads -- the objective row: pricing (:71-79), phase 1's ratios (:123-134) -- is read from the
// replica d.obj[pbuf] while workgroup 0 writes the new one to d.obj[pbuf ^ 1] (and in place).  NT: non-temporal row
// loads as well as stores (tableaux beyond the Infinity Cache: +0.6 TB/s on the bare sweep).
// ------------------------------------------------------------------------------------------
template <int T, int J, bool INPL = false, bool NT = false>
__global__ __launch_bounds__(T) void wide_kernel(Desc d, int parity, int mode, int force, const double *gather) {
    constexpr int JC = J < 4 ? J : 4; // 16-byte units per lane in flight in the pivot-row / objective-row passes
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double wd_dyn[]; // prow[pitch], lav[rpw], rhsv[rpw]; in place also colv[rpw], rin[rpw]

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING || Sin->pause) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch;
    const int rpw = (d.hcap + NB - 1) / NB; // rows per workgroup (capacity)
    double *prow = wd_dyn, *lav = wd_dyn + pitch, *rhsv = lav + rpw;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int64_t pivots_in = Sin->pivots, hist_len_in = Sin->hist_len;
    const bool swapper = b == 0 && tid == 0 && Sin->swap_valid;
    const int sw_row = Sin->swap_row, sw_col = Sin->swap_col;
    int sw_leaving = 0, sw_entering = 0;
    bool swapped = false;
    if (swapper) {
        sw_leaving = d.var[d.w + sw_row];
        sw_entering = d.var[sw_col];
    }
    auto apply_swap = [&]() __attribute__((always_inline)) {
        if (swapper && !swapped) {
            d.var[d.w + sw_row] = sw_entering;
            d.var[sw_col] = sw_leaving;
            d.pos[sw_leaving] = sw_col;
            d.pos[sw_entering] = d.w + sw_row;
        }
        swapped = true;
    };
    auto write_state = [&](int status, int phase_, int la_, int pbuf_, int mbuf_, int swap_valid_, int swap_row_,
                           int swap_col_, int64_t hist_len_, double iter_, double result_, int64_t pivots_)
                           __attribute__((always_inline)) {
        Sout->status = status;
        Sout->phase = phase_;
        Sout->bootstrap = 0;
        Sout->la = la_;
        Sout->pbuf = pbuf_;
        Sout->mbuf = mbuf_;
        Sout->pause = 0;
        Sout->dec_valid = 0;
        Sout->dec_row = 0;
        Sout->dec_col = 0;
        Sout->swap_valid = swap_valid_;
        Sout->swap_row = swap_row_;
        Sout->swap_col = swap_col_;
        Sout->pad_ = 0;
        Sout->hist_len = hist_len_;
        Sout->iter = iter_;
        Sout->result = result_;
        Sout->pivots = pivots_;
    };
    const int pbuf = mode == MODE_FUSED ? parity : Sin->pbuf;
    const int mbuf = mode == MODE_FUSED ? parity : Sin->mbuf;
    const int la_in = Sin->la;
    const double *matA = d.mat[mbuf];
    const double *rhsA = d.rhs[mbuf];
    double *matB = d.mat[INPL ? mbuf : mbuf ^ 1];
    double *rhsB = d.rhs[INPL ? mbuf : mbuf ^ 1];
    // the objective row as every workgroup reads it (in place: its replica of this parity; workgroup 0 writes the other)
    const double *__restrict__ objA = INPL ? d.obj[pbuf] : matA;
    double *objB = INPL ? d.obj[pbuf ^ 1] : nullptr;
    const bool bootstrap = Sin->bootstrap != 0;
    const int phase_in = Sin->phase;
    const double iter_in = Sin->iter;
    int phase = phase_in;
    double iter = iter_in;
    bool phase_switched = false;
    int slot = 0;

    const int gstride = SHARD_HDR + 2 * pitch;
    const int ncand = mode == MODE_SHARD ? d.nshards : NB;
    Part p_rhs, p_ratio;
    if (mode == MODE_SHARD) {
        const double *slot_ = gather + (size_t)(tid < ncand ? tid : 0) * gstride;
        p_ratio.key = slot_[0];
        p_ratio.idx = (int)slot_[1];
        p_rhs.key = slot_[2];
        p_rhs.idx = (int)slot_[3];
    } else {
        const int pi = tid < NB ? tid : 0;
        p_rhs = d.part_rhs[pbuf][pi];
        p_ratio = d.part_ratio[pbuf][pi];
    }
    auto owner_slot = [&](int grow) __attribute__((always_inline)) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < MAX_SHARDS; k++)
            if (k < d.nshards && grow >= d.bounds[k]) g = k;
        return gather + (size_t)g * gstride;
    };

    // In place the rows a workgroup sweeps do not depend on the decision: the first one is on its way while the serial
    // head of the launch (decide, pivot row, pricing: ~15 us in which nothing else streams) runs.
    [[maybe_unused]] double2 xa0[J];
    if constexpr (INPL) {
        if (mode == MODE_SHARD && !bootstrap) {
            const double *m0 = matA + (size_t)(b < h ? b : 0) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T), cs = c0 < pitch ? c0 : 0;
                xa0[j] = ld_row(m0 + cs, NT);
            }
        }
    }
    // ---------------- decide ------------------------------------------------------------------
    int row = 0, col = 0;
    bool have_pivot = false;
    if (mode != MODE_APPLY && !bootstrap) {
        int term = RUNNING;
        double term_result = NAN;
        for (;;) {
            if (!(iter < max_pivots)) {
                term = YALPS_CYCLED;
                break;
            }
            if (phase == 1) {
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_rhs.key;
                    c.i = p_rhs.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) {
                    phase = 2;
                    iter = 0.0;
                    phase_switched = true;
                    continue;
                }
                row = c.i;
                const double *mrow1 = mode == MODE_SHARD ? owner_slot(row) + SHARD_HDR + pitch : matA + (size_t)row * pitch;
                KI e = {INFINITY, INT_MAX};
#pragma unroll 1
                for (int jb = 0; jb < J; jb += JC) { // src/simplex.ts:123-134, JC 16-byte units of both rows in flight per lane
                    double2 cr[JC], ob[JC];
#pragma unroll
                    for (int j = 0; j < JC; j++) {
                        const int c0 = 2 * (tid + (jb + j) * T), cs = c0 < pitch ? c0 : 0;
                        cr[j] = *reinterpret_cast<const double2 *>(mrow1 + cs);
                        ob[j] = *reinterpret_cast<const double2 *>(objA + cs);
                    }
#pragma unroll
                    for (int j = 0; j < JC; j++) {
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int cc = 2 * (tid + (jb + j) * T) + k;
                            const double coefficient = k ? cr[j].y : cr[j].x;
                            if (cc < n && coefficient < -precision) {
                                const double ratio = -(k ? ob[j].y : ob[j].x) / coefficient;
                                if (ratio > -INFINITY && ki_better(-ratio, cc + 1, e.k, e.i)) {
                                    e.k = -ratio;
                                    e.i = cc + 1;
                                }
                            }
                        }
                    }
                }
                e = block_argmin<T>(e, sk, si, slot);
                slot ^= 1;
                if (e.i == INT_MAX) {
                    term = YALPS_INFEASIBLE;
                    break;
                }
                col = e.i;
                break;
            } else {
                col = la_in;
                if (col == 0) {
                    term = YALPS_OPTIMAL;
                    term_result = round_to_precision(rhsA[0], precision);
                    break;
                }
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_ratio.key;
                    c.i = p_ratio.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) {
                    term = YALPS_UNBOUNDED;
                    term_result = (double)col;
                    break;
                }
                row = c.i;
                break;
            }
        }
        if (term != RUNNING) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(term, phase, la_in, pbuf, mbuf, 0, 0, 0, phase_switched ? 0 : hist_len_in, iter, term_result,
                            pivots_in);
            return;
        }
        if (mode == MODE_SHARD && C->check_cycles && d.cyc_verdict[parity & 1]) { // :98,137: shard_cycle_kernel's verdict on this pivot
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(YALPS_CYCLED, phase, la_in, pbuf, mbuf, 0, 0, 0, (phase_switched ? 0 : hist_len_in) + 1, iter, NAN, pivots_in);
            return;
        }
        have_pivot = true;
    } else if (mode == MODE_APPLY && Sin->dec_valid) {
        row = Sin->dec_row;
        col = Sin->dec_col;
        have_pivot = true;
    }
    if (!have_pivot && !bootstrap) {
        apply_swap();
        if (b == 0 && tid == 0 && !(force & 1))
            write_state(RUNNING, phase_in, la_in, pbuf, mbuf, 0, 0, 0, hist_len_in, iter_in, NAN, pivots_in);
        return;
    }

    // ---------------- prepare: normalised pivot row -> LDS, look-ahead pricing --------------------
    const int colx = have_pivot ? col - 1 : 0;
    const double *gslot = mode == MODE_SHARD ? owner_slot(row) : nullptr;
    const double *mrow = mode == MODE_SHARD ? gslot + SHARD_HDR + (phase == 1 ? pitch : 0) : matA + (size_t)row * pitch;
    const int lrow = !have_pivot ? -1
                     : mode != MODE_SHARD ? row
                     : (row >= d.bounds[d.shard_rank] && row < d.bounds[d.shard_rank + 1]) ? row - d.row_base : -1;
    const double q = have_pivot ? mrow[colx] : 1.0;
    const double coef0 = have_pivot ? objA[colx] : 0.0;
    const double rhs_row = have_pivot ? (mode == MODE_SHARD ? gslot[phase == 1 ? 5 : 4] : rhsA[row]) : 0.0;
    const double inv_q = 1.0 / q;
    const double flushed = __longlong_as_double((long long)FLUSHED);
    // (a lane normalises, prices and later sweeps the same J 16-byte units of every row: all of a lane's loads of a row are
    // in flight at once -- as run-time loops of 8-byte accesses these two passes were ~30 dependent round trips at 16385
    // columns, during which no row streamed)
    int la = 0;
    {
        const bool touched0 = have_pivot && fabs(coef0) > 1e-16;
        const double nq0 = -coef0 / q; // :36 for the objective row
        KI best = {INFINITY, INT_MAX};
#pragma unroll 1
        for (int jb = 0; jb < J; jb += JC) {
        double2 pvr[JC], ob[JC];
#pragma unroll
        for (int j = 0; j < JC; j++) {
            const int c0 = 2 * (tid + (jb + j) * T), cs = c0 < pitch ? c0 : 0;
            pvr[j] = have_pivot ? *reinterpret_cast<const double2 *>(mrow + cs) : make_double2(0.0, 0.0);
            ob[j] = *reinterpret_cast<const double2 *>(objA + cs);
        }
#pragma unroll
        for (int j = 0; j < JC; j++) {
            const int c0 = 2 * (tid + (jb + j) * T);
            if (c0 >= pitch) continue;
            double2 v = pvr[j];
            v.x = fabs(v.x) > 1e-16 ? v.x / q : flushed;
            v.y = fabs(v.y) > 1e-16 ? v.y / q : flushed;
            *reinterpret_cast<double2 *>(prow + c0) = v;
#pragma unroll
            for (int k = 0; k < 2; k++) { // Dantzig pricing of the objective row as it is after this pivot (:71-79)
                const int cc = c0 + k;
                double ov = k ? ob[j].y : ob[j].x;
                if (touched0) {
                    const double pn = k ? v.y : v.x;
                    if (cc == colx)
                        ov = nq0;
                    else if ((unsigned long long)__double_as_longlong(pn) != FLUSHED) {
                        const double prod = coef0 * pn;
                        ov = ov - prod;
                    }
                }
                if (cc < n && ov > precision && ki_better(-ov, cc + 1, best.k, best.i)) {
                    best.k = -ov;
                    best.i = cc + 1;
                }
            }
        }
        }
        best = block_argmin<T>(best, sk, si, slot); // (its barrier also publishes prow)
        slot ^= 1;
        la = best.i == INT_MAX ? 0 : best.i;
    }
    const int lax = la > 0 ? la - 1 : -1; // mat column of the next entering variable
    const bool nz_rhs = fabs(rhs_row) > 1e-16;
    const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;

    // ---------------- body: stream my rows, one at a time, double-buffered ----------------------
    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    double *colv = rhsv + rpw, *rin = colv + rpw; // (in place only) my rows' pivot-column entries and RHS as they are before this pivot
    if constexpr (INPL) {
        for (int i = tid; i < my_rows; i += T) {
            const int r = b + NB * i;
            colv[i] = matA[(size_t)r * pitch + colx];
            rin[i] = rhsA[r];
        }
        __syncthreads();
    }
    auto load_row = [&](double2 (&x)[J], double &cf, double &rr, int i) __attribute__((always_inline)) {
        const int r = b + NB * i;
        const int rs = r < h ? r : b; // in-bounds dummy past the end (b < h whenever this is reached)
        if constexpr (INPL) {
            cf = colv[r < h ? i : 0];
            rr = rin[r < h ? i : 0];
        } else {
            cf = matA[(size_t)rs * pitch + colx];
            rr = rhsA[rs];
        }
        const double *mr = matA + (size_t)rs * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) x[j] = ld_row(mr + cofs[j], NT);
    };
    auto process = [&](double2 (&x)[J], double cf, double rr, int i) __attribute__((always_inline)) {
        const int r = b + NB * i;
        if (r >= h) return;
        double my_rhs = rr;
        const bool is_pivot_row = have_pivot && r == lrow;
        const bool act = have_pivot && !is_pivot_row && fabs(cf) > 1e-16; // src/simplex.ts:31
        const double nq = act ? -cf / q : 0.0;                            // :36 (uniform)
        if (is_pivot_row)
            my_rhs = pn_rhs;
        else if (act && nz_rhs) {
            const double prod = cf * pn_rhs;
            my_rhs = rr - prod;
        }
        double *mr = matB + (size_t)r * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (c0 >= pitch) continue;
            double2 v = x[j];
            if (is_pivot_row || act) {
                const double2 pn = *reinterpret_cast<const double2 *>(prow + c0);
                const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
                const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
                if (is_pivot_row) {
                    v.x = f0 ? pn.x : 0.0;
                    v.y = f1 ? pn.y : 0.0;
                    if (c0 == colx) v.x = inv_q; // :25
                    if (c0 + 1 == colx) v.y = inv_q;
                } else {
                    const double px = cf * pn.x, py = cf * pn.y;
                    const double nx = v.x - px, ny = v.y - py;
                    v.x = f0 ? nx : v.x;
                    v.y = f1 ? ny : v.y;
                    if (c0 == colx) v.x = nq;
                    if (c0 + 1 == colx) v.y = nq;
                }
            }
            if (c0 == (lax & ~1)) lav[i] = (lax & 1) ? v.y : v.x; // the row's entry in the next entering column
            if constexpr (INPL) {
                if (r == 0) *reinterpret_cast<double2 *>(objB + c0) = v; // (workgroup 0: the next launch's replica, changed or not)
                if (!(is_pivot_row || act)) continue;                     // in place: an untouched row stays where it is
            }
            if (force & 64) {
                st_row_nt(mr + c0, v);
            } else
                *reinterpret_cast<double2 *>(mr + c0) = v;
        }
        if (tid == 0) {
            rhsB[r] = my_rhs;
            rhsv[i] = my_rhs;
        }
    };
    if (my_rows > 0) {
        double2 xa[J], xb[J];
        double cfa, cfb, rra, rrb;
        const bool preloaded = INPL && mode == MODE_SHARD && !bootstrap; // (uniform)
        if (preloaded) {
#pragma unroll
            for (int j = 0; j < J; j++) xa[j] = xa0[j];
            cfa = colv[0];
            rra = rin[0];
        } else {
            load_row(xa, cfa, rra, 0);
        }
        for (int i = 0; i < my_rows; i += 2) {
            load_row(xb, cfb, rrb, i + 1);
            process(xa, cfa, rra, i);
            load_row(xa, cfa, rra, i + 2);
            process(xb, cfb, rrb, i + 1);
        }
    }
    __syncthreads(); // lav / rhsv complete
    KI cand_ratio = {INFINITY, INT_MAX}, cand_rhs = {INFINITY, INT_MAX};
    for (int i = tid; i < my_rows; i += T) {
        const int r = b + NB * i;
        if (r < 1) continue;
        const int gr = r + d.row_base;
        const double my_rhs = rhsv[i];
        if (my_rhs < -precision && ki_better(my_rhs, gr, cand_rhs.k, cand_rhs.i)) {
            cand_rhs.k = my_rhs;
            cand_rhs.i = gr;
        }
        if (la > 0) {
            const double value = lav[i];
            if (value > precision) {
                const double ratio = my_rhs / value;
                if (ratio < INFINITY) {
                    const double key = (ratio <= precision) ? -INFINITY : ratio;
                    if (ki_better(key, gr, cand_ratio.k, cand_ratio.i)) {
                        cand_ratio.k = key;
                        cand_ratio.i = gr;
                    }
                }
            }
        }
    }
    cand_ratio = block_argmin<T>(cand_ratio, sk, si, slot);
    slot ^= 1;
    cand_rhs = block_argmin<T>(cand_rhs, sk, si, slot);
    slot ^= 1;
    if (tid == 0) {
        Part p;
        p.pad_ = 0;
        p.key = cand_ratio.k;
        p.idx = cand_ratio.i;
        d.part_ratio[pbuf ^ 1][b] = p;
        p.key = cand_rhs.k;
        p.idx = cand_rhs.i;
        d.part_rhs[pbuf ^ 1][b] = p;
    }
    apply_swap();
    if (b == 0 && tid == 0 && !(force & 1)) {
        const bool counted = have_pivot && mode != MODE_APPLY;
        write_state(RUNNING, phase, la, pbuf ^ 1, INPL ? mbuf : mbuf ^ 1, have_pivot ? 1 : 0, row, col,
                    ((mode != MODE_APPLY && phase_switched) ? 0 : hist_len_in) + (mode == MODE_SHARD && have_pivot && C->check_cycles ? 1 : 0),
                    counted ? iter + 1.0 : iter, NAN,
                    counted ? pivots_in + 1 : pivots_in);
    }
}
