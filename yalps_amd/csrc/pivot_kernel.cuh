// pivot_kernel.cuh -- streaming pivot, one launch per pivot, rows batched in registers
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// pivot_kernel<T lanes, J units per lane per row, R rows per lane and batch, D rows prefetched>
//   lane `tid`, unit j  <->  mat columns 2*(tid + j*T) + {0,1}  <->  reference columns +1
//   workgroup b owns rows b, b+NB, b+2NB, ...; lanes 0..R-1 of wave 0 also own the scalar side
//   (RHS entry, pivot-column entry, ratio) of row g = lane.
// Load discipline: every load is unconditional with an in-bounds (possibly dummy) address and
// the body is fully unrolled, so hipcc can count the load queue (s_waitcnt vmcnt(N)) instead of
// draining it; vmcnt retires in issue order, hence the issue order below is deliberate.
// ------------------------------------------------------------------------------------------
template <int T, int J, int R, int D>
__global__ __launch_bounds__(T) void pivot_kernel(Desc d, int parity, int mode, int force, const double *gather) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_la[2][R]; // next entering column's entries of my rows (ping-pong per batch)
    __shared__ int cyc_flag;

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING || Sin->pause) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int64_t pivots_in = Sin->pivots, hist_len_in = Sin->hist_len;
    // (0) pending basis bookkeeping of the previous pivot: its two loads are the oldest of this
    // launch, its four stores are fire-and-forget at the end (or before any early return)
    const bool swapper = b == 0 && tid == 0 && Sin->swap_valid;
    const int sw_row = Sin->swap_row, sw_col = Sin->swap_col;
    int sw_leaving = 0, sw_entering = 0;
    bool swapped = false;
    if (swapper) {
        sw_leaving = d.var[d.w + sw_row];
        sw_entering = d.var[sw_col];
    }
    auto apply_swap = [&]() {
        if (swapper && !swapped) {
            d.var[d.w + sw_row] = sw_entering;
            d.var[sw_col] = sw_leaving;
            d.pos[sw_leaving] = sw_col;
            d.pos[sw_entering] = d.w + sw_row;
        }
        swapped = true;
    };
    // every field of the next state, from registers
    auto write_state = [&](int status, int phase_, int bootstrap_, int la_, int pbuf_, int mbuf_, int pause_,
                           int dec_valid_, int dec_row_, int dec_col_, int swap_valid_, int swap_row_,
                           int swap_col_, int64_t hist_len_, double iter_, double result_, int64_t pivots_) {
        Sout->status = status;
        Sout->phase = phase_;
        Sout->bootstrap = bootstrap_;
        Sout->la = la_;
        Sout->pbuf = pbuf_;
        Sout->mbuf = mbuf_;
        Sout->pause = pause_;
        Sout->dec_valid = dec_valid_;
        Sout->dec_row = dec_row_;
        Sout->dec_col = dec_col_;
        Sout->swap_valid = swap_valid_;
        Sout->swap_row = swap_row_;
        Sout->swap_col = swap_col_;
        Sout->pad_ = 0;
        Sout->hist_len = hist_len_;
        Sout->iter = iter_;
        Sout->result = result_;
        Sout->pivots = pivots_;
    };
    // Every launch that gets past the selection flips both ping-pong indices, so in FUSED graphs
    // they equal the launch parity (a kernel argument): the first loads need not wait for the state.
    const int pbuf = mode == MODE_FUSED ? parity : Sin->pbuf;
    const int mbuf = mode == MODE_FUSED ? parity : Sin->mbuf;
    const int la_in = Sin->la;
    const double *__restrict__ matA = d.mat[mbuf];
    const double *__restrict__ rhsA = d.rhs[mbuf];
    double *__restrict__ matB = d.mat[mbuf ^ 1];
    double *__restrict__ rhsB = d.rhs[mbuf ^ 1];
    const bool bootstrap = Sin->bootstrap != 0;
    const int phase_in = Sin->phase;
    const double iter_in = Sin->iter;
    int phase = phase_in;
    double iter = iter_in;
    bool phase_switched = false;
    int slot = 0; // block_argmin scratch ping-pong

    // lane's column offsets (lanes past the row end use column 0: in-bounds dummy)
    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }

    // (1) control loads first (oldest in the queue): partials of the previous launch, objective row
    // (SHARD mode: the all-gathered per-rank candidates instead -- slot layout at shard_select_kernel)
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
    // where a (global) row's raw data and RHS entry come from: my tableau, or its owner's gather slot
    auto owner_slot = [&](int grow) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < MAX_SHARDS; k++)
            if (k < d.nshards && grow >= d.bounds[k]) g = k;
        return gather + (size_t)g * gstride;
    };
    double2 o[J]; // objective row slice (reduced costs)
#pragma unroll
    for (int j = 0; j < J; j++) o[j] = *reinterpret_cast<const double2 *>(matA + cofs[j]);

    // (2) the first D rows of my first batch: they depend on nothing the selection decides, so
    // they stream in while the selection runs
    double2 x[R][J];
    {
#pragma unroll
        for (int g = 0; g < D; g++) {
            const int r = b + NB * g;
            const double *mr = matA + (size_t)(r < h ? r : b) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) x[g][j] = ld_row(mr + cofs[j], false);
        }
    }

    int row = 0, col = 0;
    bool have_pivot = false, pv_loaded = false;
    double2 pv[J]; // pivot row slice: raw, then normalised
#pragma unroll
    for (int j = 0; j < J; j++) pv[j] = make_double2(0.0, 0.0);

    // ---------------- decide: which pivot, or stop (src/simplex.ts:66-142 minus pivot()) ------
    if (mode != MODE_APPLY && !bootstrap) {
        int term = RUNNING;
        double term_result = NAN;
        for (;;) {
            if (!(iter < max_pivots)) { // loop bounds :69,109 -> "cycled" :102,141
                term = YALPS_CYCLED;
                break;
            }
            if (phase == 1) {
                // leaving row: most negative RHS, strict <, first wins (:111-119)
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_rhs.key;
                    c.i = p_rhs.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) { // :120 tail call of phase2: fresh counter and history
                    phase = 2;
                    iter = 0.0;
                    phase_switched = true;
                    continue;
                }
                row = c.i;
                // entering column: max -M[0,c]/M[row,c] over M[row,c] < -precision (:123-134)
                const double *mrow = mode == MODE_SHARD ? owner_slot(row) + SHARD_HDR + pitch
                                                       : matA + (size_t)row * pitch;
                KI e = {INFINITY, INT_MAX};
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    pv[j] = *reinterpret_cast<const double2 *>(mrow + cofs[j]);
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
                pv_loaded = true;
                e = block_argmin<T>(e, sk, si, slot);
                slot ^= 1;
                if (e.i == INT_MAX) { // :135
                    term = YALPS_INFEASIBLE;
                    break;
                }
                col = e.i;
                break;
            } else {
                col = la_in; // Dantzig pricing (:71-79) was done by the previous launch
                if (col == 0) { // :80
                    term = YALPS_OPTIMAL;
                    term_result = round_to_precision(rhsA[0], precision);
                    break;
                }
                // leaving row: min-ratio test with the early break (:83-95); the partials carry
                // key = -inf for "ratio <= precision" rows so the lowest such index wins
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_ratio.key;
                    c.i = p_ratio.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) { // :96
                    term = YALPS_UNBOUNDED;
                    term_result = (double)col;
                    break;
                }
                row = c.i;
                break;
            }
        }
        int64_t hist_len = phase_switched ? 0 : hist_len_in;
        if (term != RUNNING) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(term, phase, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len, iter, term_result,
                            pivots_in);
            return;
        }
        if (C->check_cycles && mode == MODE_SHARD) { // :98,137: shard_cycle_kernel has run the detector on this very pivot
            hist_len += 1;
            if (d.cyc_verdict[parity & 1]) {
                apply_swap();
                if (b == 0 && tid == 0)
                    write_state(YALPS_CYCLED, phase, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len, iter, NAN, pivots_in);
                return;
            }
        } else if (C->check_cycles) { // :98,137 (DECIDE launches only: one workgroup)
            if (hist_len >= C->hist_cap) { // history full: the host grows it; nothing is consumed
                apply_swap();
                if (tid == 0)
                    write_state(RUNNING, phase, 0, la_in, pbuf, mbuf, 1, 0, 0, 0, 0, 0, 0, hist_len, iter, NAN,
                                pivots_in);
                return;
            }
            apply_swap(); // the detector reads the basis as it is now
            __syncthreads();
            const bool cyc = has_cycle(C, hist_len, d.var[d.w + row], d.var[col], &cyc_flag);
            hist_len += 1;
            if (cyc) {
                if (tid == 0)
                    write_state(YALPS_CYCLED, phase, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len, iter, NAN,
                                pivots_in);
                return;
            }
        }
        have_pivot = true;
        if (mode == MODE_DECIDE) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(RUNNING, phase, 0, la_in, pbuf, mbuf, 0, 1, row, col, 0, 0, 0, hist_len, iter + 1.0, NAN,
                            pivots_in + 1);
            return;
        }
    } else if (mode == MODE_APPLY && Sin->dec_valid) {
        row = Sin->dec_row;
        col = Sin->dec_col;
        have_pivot = true;
    }
    if (!have_pivot && !bootstrap) { // APPLY with nothing decided
        apply_swap();
        if (b == 0 && tid == 0 && !(force & 1))
            write_state(RUNNING, phase_in, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len_in, iter_in, NAN,
                        pivots_in);
        return;
    }

    // ---------------- prepare: pivot row normalise + look-ahead pricing -----------------------
    // owner lane/unit/element of a reference column c (c >= 1): mat column c-1
    const int ucol = (col - 1) >> 1, ecol = (col - 1) & 1;
    const int col_tid = have_pivot ? ucol % T : -1, col_j = have_pivot ? ucol / T : -1;
    const int colx = have_pivot ? col - 1 : 0; // in-bounds even without a pivot
    // (3) pivot row, quotient, objective row's pivot-column entry
    // (SHARD: the pivot row travels in its owner's gather slot -- the ratio candidate's row in
    // phase 2, the most-negative-RHS candidate's row in phase 1; `row` is a GLOBAL index and
    // `lrow` its local index here, -1 if another rank owns it)
    const double *gslot = mode == MODE_SHARD ? owner_slot(row) : nullptr;
    const double *mrow = mode == MODE_SHARD ? gslot + SHARD_HDR + (phase == 1 ? pitch : 0) : matA + (size_t)row * pitch;
    const int lrow = !have_pivot ? -1
                     : mode != MODE_SHARD ? row
                     : (row >= d.bounds[d.shard_rank] && row < d.bounds[d.shard_rank + 1]) ? row - d.row_base : -1;
    if (!pv_loaded) {
#pragma unroll
        for (int j = 0; j < J; j++) pv[j] = *reinterpret_cast<const double2 *>(mrow + cofs[j]);
    }
    const double q_ld = mrow[colx], coef0_ld = matA[colx];
    const double rhs_row = mode == MODE_SHARD ? gslot[phase == 1 ? 5 : 4] : rhsA[row];
    const double q = have_pivot ? q_ld : 1.0, coef0 = have_pivot ? coef0_ld : 0.0;

    unsigned nzmask = 0; // bit (2j+k): pivot-row entry is in nonZeroColumns (:18-23)
    if (have_pivot) {    // src/simplex.ts:14-25
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double v = elem(pv[j], k);
                const bool nz = fabs(v) > 1e-16;
                pv[j] = with_elem(pv[j], k, nz ? v / q : 0.0);
                if (nz) nzmask |= 1u << (2 * j + k);
            }
        }
    }
    const double inv_q = 1.0 / q; // :25 (the pivot entry becomes 1/quotient)
    int la = 0, la_tid = -1, la_j = -1, ela = 0;
    bool la_known = false;

    // ---------------- body: eliminate my rows into the other buffer, emit partials ------------
    KI cand_ratio = {INFINITY, INT_MAX}, cand_rhs = {INFINITY, INT_MAX}; // lanes 0..R-1
    for (int i0 = 0; b + NB * i0 < h; i0 += R) {
        const int r_first = b + NB * i0;
        // (4) pivot-column entries of my rows (uniform per row) and, lane g, the RHS of row g
        double coef[R];
#pragma unroll
        for (int g = 0; g < R; g++) {
            const int r = b + NB * (i0 + g);
            coef[g] = matA[(size_t)(r < h ? r : r_first) * pitch + colx];
        }
        const int my_r = b + NB * (i0 + tid);
        const bool my_live = tid < R && my_r < h;
        const double rr = rhsA[my_live ? my_r : 0];
        if (i0 > 0) { // later batches: prefetch their first D rows (batch 0's came in at the top)
#pragma unroll
            for (int g = 0; g < D; g++) {
                const int r = b + NB * (i0 + g);
                const double *mr = matA + (size_t)(r < h ? r : r_first) * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) x[g][j] = ld_row(mr + cofs[j], false);
            }
        }
        // lane g: scalar side of row g -- RHS entry (:33 at c = 0) and pivot-column entry (:36)
        double my_rhs = rr, my_val = 0.0;
        bool my_val_set = false;
        if (my_live && have_pivot) {
            double my_coef = 0.0;
#pragma unroll
            for (int g = 0; g < R; g++)
                if (tid == g) my_coef = coef[g];
            const bool nz_rhs = fabs(rhs_row) > 1e-16;
            const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
            if (my_r == lrow) {
                my_rhs = pn_rhs;
            } else if (fabs(my_coef) > 1e-16) {
                if (nz_rhs) {
                    const double prod = my_coef * pn_rhs;
                    my_rhs = rr - prod;
                }
                const double nq = -my_coef / q;
                matB[(size_t)my_r * pitch + colx] = nq; // the owner lane stores only the other half
                my_val = nq; // what this row holds in column `col` from now on
                my_val_set = true;
            }
            rhsB[my_r] = my_rhs;
        } else if (my_live) {
            rhsB[my_r] = rr; // bootstrap: carry over
        }
        // rows: eliminate + write to the other buffer, row by row, with the loads of the next
        // rows in flight.  Rows the reference leaves untouched (:31) are carried over unchanged.
#pragma unroll
        for (int g = 0; g < R; g++) {
            const int r = b + NB * (i0 + g);
            const bool live = r < h;
            const double c = coef[g];
            const bool act = have_pivot && live && r != lrow && fabs(c) > 1e-16;
            if (have_pivot && live && r == lrow) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    x[g][j] = pv[j];
                    if (tid == col_tid && j == col_j) x[g][j] = with_elem(x[g][j], ecol, inv_q);
                }
            } else if (act) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    if (nzmask & (1u << (2 * j))) {
                        const double prod = c * pv[j].x;
                        x[g][j].x = x[g][j].x - prod;
                    }
                    if (nzmask & (1u << (2 * j + 1))) {
                        const double prod = c * pv[j].y;
                        x[g][j].y = x[g][j].y - prod;
                    }
                }
            }
            if (live) { // (a bootstrap launch just carries the tableau over)
                double *mr = matB + (size_t)r * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 >= pitch) continue;
                    if (act && tid == col_tid && j == col_j) // lane g of wave 0 stores -coef/quotient
                        mr[c0 + (ecol ^ 1)] = elem(x[g][j], ecol ^ 1);
                    else if (force & 64) { // streaming (non-temporal) stores; YALPS_HIP_NT=0 turns them off
                        st_row_nt(mr + c0, x[g][j]);
                    } else
                        *reinterpret_cast<double2 *>(mr + c0) = x[g][j];
                }
            }
            if (g + D < R) { // keep D rows in flight
                const int rn = b + NB * (i0 + g + D);
                const double *mr = matA + (size_t)(rn < h ? rn : r_first) * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) x[g + D][j] = ld_row(mr + cofs[j], false);
            }
        }
        // Dantzig pricing (:71-79) of the objective row as it is AFTER this pivot (as it is, when
        // bootstrapping) -> entering column `la` of the next iteration
        if (!la_known) {
            la_known = true;
            const bool touched0 = have_pivot && fabs(coef0) > 1e-16;
            KI best = {INFINITY, INT_MAX};
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    double ov = elem(o[j], k);
                    if (touched0) {
                        if (tid == col_tid && j == col_j && k == ecol)
                            ov = -coef0 / q;
                        else if (nzmask & (1u << (2 * j + k))) {
                            const double prod = coef0 * elem(pv[j], k);
                            ov = ov - prod;
                        }
                    }
                    if (c0 + k < n && ov > precision && ki_better(-ov, c0 + k + 1, best.k, best.i)) {
                        best.k = -ov;
                        best.i = c0 + k + 1;
                    }
                }
            }
            best = block_argmin<T>(best, sk, si, slot);
            slot ^= 1;
            la = best.i == INT_MAX ? 0 : best.i;
            const int ula = (la - 1) >> 1;
            ela = (la - 1) & 1;
            la_tid = la > 0 ? ula % T : -1;
            la_j = la > 0 ? ula / T : -1;
        }
        // entries of my rows in column `la` (x holds the rows as written), for lanes 0..R-1
        if (tid == la_tid) {
#pragma unroll
            for (int g = 0; g < R; g++)
#pragma unroll
                for (int j = 0; j < J; j++)
                    if (j == la_j) sh_la[(i0 / R) & 1][g] = elem(x[g][j], ela);
        }
        __syncthreads(); // sh_la of this batch visible to lanes 0..R-1
        // candidates of my row for the next launch's scans
        if (my_live && my_r >= 1) {
            const int my_gr = my_r + d.row_base; // global row index
            if (my_rhs < -precision && ki_better(my_rhs, my_gr, cand_rhs.k, cand_rhs.i)) {
                cand_rhs.k = my_rhs;
                cand_rhs.i = my_gr;
            }
            if (la > 0) {
                const double value = (my_val_set && la == col) ? my_val : sh_la[(i0 / R) & 1][tid];
                if (value > precision) {
                    const double ratio = my_rhs / value;
                    if (ratio < INFINITY) {
                        const double key = (ratio <= precision) ? -INFINITY : ratio;
                        if (ki_better(key, my_gr, cand_ratio.k, cand_ratio.i)) {
                            cand_ratio.k = key;
                            cand_ratio.i = my_gr;
                        }
                    }
                }
            }
        }
    }
    if (tid < 64) { // R <= 16 < 64: wave 0 holds every candidate
        cand_ratio = wave_argmin(cand_ratio);
        cand_rhs = wave_argmin(cand_rhs);
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
    }
    apply_swap();
    if (b == 0 && tid == 0 && !(force & 1)) {
        const bool counted = have_pivot && mode != MODE_APPLY; // DECIDE already counted an APPLY's pivot
        write_state(RUNNING, phase, 0, la, pbuf ^ 1, mbuf ^ 1, 0, 0, 0, 0, have_pivot ? 1 : 0, row, col,
                    ((mode != MODE_APPLY && phase_switched) ? 0 : hist_len_in) + (mode == MODE_SHARD && have_pivot && C->check_cycles ? 1 : 0),
                    counted ? iter + 1.0 : iter, NAN,
                    counted ? pivots_in + 1 : pivots_in);
    }
}
