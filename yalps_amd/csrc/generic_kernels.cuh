// generic_kernels.cuh -- any-shape fallback: rows of any width (> 16385 columns), one DECIDE + one APPLY launch per pivot
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// The tuned kernels span a row with lanes x compile-time units (<= 16384 columns).  The reference has no such
// limit (a model with 100 000 variables and 200 constraints is a 160 MB tableau), so tableaux beyond it take this
// pair, written with run-time loops only, in place, bit-exact like everything else:
//   generic_decide_kernel  ONE workgroup: loop bound, phase-1 / phase-2 selection (src/simplex.ts:106-134,
//                          :66-96), hasCycle (:44-63), basis swap (:7-12), pivot row normalised in place and
//                          into gen_prow[] (FLUSHED marks the zeroed entries, :14-25); decision left in the state
//   generic_apply_kernel   one workgroup per CU over rows b, b + NB, ...: pivot-column entries of my rows gathered
//                          first, then only the touched rows (|coef| > 1e-16, :31) are streamed and updated
// State lives in d.st[0] (no ping-pong: the two kernels alternate on one stream).  gen_scal = {quotient, RHS of the
// normalised pivot row, 1.0 if column 0 is one of the pivot row's non-zero columns}.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void generic_decide_kernel(Desc d) {
    constexpr int T = 1024;
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ int sh_flag;
    YState *S = d.st;
    const YConst *C = d.cst;
    const int tid = threadIdx.x;
    if (S->status != RUNNING) {
        if (tid == 0) S->dec_valid = 0;
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch, w = d.w;
    const double precision = C->precision, max_pivots = C->max_pivots;
    double *mat = d.mat[0], *rhs = d.rhs[0];
    int phase = S->phase, slot = 0;
    double iter = S->iter;
    int64_t hist_len = S->hist_len;
    const int64_t pivots = S->pivots;
    __syncthreads(); // everybody has read the state before lane 0 rewrites it
    int term = RUNNING, row = 0, col = 0;
    double term_result = NAN;
    for (;;) {
        if (!(iter < max_pivots)) {
            term = YALPS_CYCLED;
            break;
        }
        if (phase == 1) {
            KI c = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += T) {
                const double v = rhs[r];
                if (v < -precision && ki_better(v, r, c.k, c.i)) {
                    c.k = v;
                    c.i = r;
                }
            }
            c = block_argmin<T>(c, sk, si, slot);
            slot ^= 1;
            if (c.i == INT_MAX) {
                phase = 2;
                iter = 0.0;
                hist_len = 0;
                continue;
            }
            row = c.i;
            const double *mrow = mat + (size_t)row * pitch;
            KI e = {INFINITY, INT_MAX};
            for (int cc = tid; cc < n; cc += T) {
                const double coefficient = mrow[cc];
                if (coefficient < -precision) {
                    const double ratio = -mat[cc] / coefficient;
                    if (ratio > -INFINITY && ki_better(-ratio, cc + 1, e.k, e.i)) {
                        e.k = -ratio;
                        e.i = cc + 1;
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
        } else {
            KI pr = {INFINITY, INT_MAX};
            for (int cc = tid; cc < n; cc += T) {
                const double rc = mat[cc];
                if (rc > precision && ki_better(-rc, cc + 1, pr.k, pr.i)) {
                    pr.k = -rc;
                    pr.i = cc + 1;
                }
            }
            pr = block_argmin<T>(pr, sk, si, slot);
            slot ^= 1;
            if (pr.i == INT_MAX) {
                term = YALPS_OPTIMAL;
                term_result = round_to_precision(rhs[0], precision);
                break;
            }
            col = pr.i;
            KI c = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += T) {
                const double value = mat[(size_t)r * pitch + col - 1];
                if (value <= precision) continue;
                const double ratio = rhs[r] / value;
                if (!(ratio < INFINITY)) continue;
                const double key = (ratio <= precision) ? -INFINITY : ratio;
                if (ki_better(key, r, c.k, c.i)) {
                    c.k = key;
                    c.i = r;
                }
            }
            c = block_argmin<T>(c, sk, si, slot);
            slot ^= 1;
            if (c.i == INT_MAX) {
                term = YALPS_UNBOUNDED;
                term_result = (double)col;
                break;
            }
            row = c.i;
        }
        if (C->check_cycles) { // :98,137 (the host keeps the history long enough for every launch of a batch)
            if (has_cycle(C, hist_len, d.var[w + row], d.var[col], &sh_flag)) {
                term = YALPS_CYCLED;
                break;
            }
            hist_len += 1;
        }
        break; // a pivot to apply
    }
    if (term != RUNNING) {
        if (tid == 0) {
            S->status = term;
            S->phase = phase;
            S->iter = iter;
            S->result = term_result;
            S->hist_len = hist_len;
            S->dec_valid = 0;
        }
        return;
    }
    // ---- pivot(row, col), the part that concerns the pivot row: src/simplex.ts:6-25 ----
    double *mrow = mat + (size_t)row * pitch;
    const double q = mrow[col - 1], rhs_row = rhs[row];
    const double flushed = __longlong_as_double((long long)FLUSHED);
    __syncthreads(); // q read by everybody before the row changes
    for (int c = tid; c < pitch; c += T) {
        const double v = mrow[c];
        const bool nz = fabs(v) > 1e-16;
        const double pn = nz ? v / q : 0.0;
        mrow[c] = (c == col - 1) ? 1.0 / q : pn;
        d.gen_prow[c] = nz ? pn : flushed;
    }
    if (tid == 0) {
        const double pn_rhs = fabs(rhs_row) > 1e-16 ? rhs_row / q : 0.0;
        rhs[row] = pn_rhs;
        d.gen_scal[0] = q;
        d.gen_scal[1] = pn_rhs;
        d.gen_scal[2] = fabs(rhs_row) > 1e-16 ? 1.0 : 0.0;
        const int leaving = d.var[w + row], entering = d.var[col]; // :7-12
        d.var[w + row] = entering;
        d.var[col] = leaving;
        d.pos[leaving] = col;
        d.pos[entering] = w + row;
        S->phase = phase;
        S->iter = iter + 1.0;
        S->pivots = pivots + 1;
        S->hist_len = hist_len;
        S->dec_row = row;
        S->dec_col = col;
        S->dec_valid = 1;
    }
}

__global__ __launch_bounds__(1024) void generic_apply_kernel(Desc d) {
    constexpr int T = 1024;
    extern __shared__ double ga_colv[]; // pivot-column entries of my rows
    const YState *S = d.st;
    if (!S->dec_valid) return;
    const YConst *C = d.cst;
    const int tid = threadIdx.x, NB = gridDim.x, b = blockIdx.x;
    const int h = C->height, pitch = d.pitch, row = S->dec_row, colx = S->dec_col - 1, units = pitch / 2;
    double *mat = d.mat[0], *rhs = d.rhs[0];
    const double q = d.gen_scal[0], pn_rhs = d.gen_scal[1];
    const bool nz_rhs = d.gen_scal[2] != 0.0;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    for (int i = tid; i < my_rows; i += T) {
        const int r = b + NB * i;
        ga_colv[i] = r == row ? 0.0 : mat[(size_t)r * pitch + colx]; // (the pivot row is done; 0.0 = not touched)
    }
    __syncthreads();
    for (int i = tid; i < my_rows; i += T) { // RHS entries (:33 at column 0)
        const double coef = ga_colv[i];
        if (fabs(coef) > 1e-16 && nz_rhs) {
            const int r = b + NB * i;
            const double prod = coef * pn_rhs;
            rhs[r] = rhs[r] - prod;
        }
    }
    for (int i = 0; i < my_rows; i++) {
        const double coef = ga_colv[i];
        if (!(fabs(coef) > 1e-16)) continue; // :31 (uniform)
        double *mr = mat + (size_t)(b + NB * i) * pitch;
        const double nq = -coef / q;
        for (int u = tid; u < units; u += T) {
            double2 x = *reinterpret_cast<const double2 *>(mr + 2 * u);
            const double2 p = *reinterpret_cast<const double2 *>(d.gen_prow + 2 * u);
            if ((unsigned long long)__double_as_longlong(p.x) != FLUSHED) {
                const double prod = coef * p.x;
                x.x = x.x - prod;
            }
            if ((unsigned long long)__double_as_longlong(p.y) != FLUSHED) {
                const double prod = coef * p.y;
                x.y = x.y - prod;
            }
            if (u == (colx >> 1)) {
                if (colx & 1)
                    x.y = nq;
                else
                    x.x = nq;
            }
            *reinterpret_cast<double2 *>(mr + 2 * u) = x;
        }
    }
}
