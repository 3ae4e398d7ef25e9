// batch_kernel.cuh -- batched branch-and-cut nodes, one workgroup per node
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// batch_kernel: many independent branch-and-cut nodes at once, ONE workgroup per node
// (BASELINE config 4).  A node's LP is the root's optimal tableau plus one
// row per cut (src/branchAndCut.ts:22-61 `applyCuts`), re-solved with simplex() (:127).  The
// root stays resident in HBM; every workgroup builds its node's tableau in its own workspace and
// runs the whole two-phase loop there with workgroup-local synchronisation only -- nodes are
// independent, there is no cross-workgroup communication and no host round trip.
// Per pivot: the pivot column is gathered into LDS first (so rows can then be updated in place),
// the pivot row is normalised into LDS (FLUSHED marks entries pivot() zeroed), the sweep gives
// every lane fixed 16-byte column units and walks the rows.
// ------------------------------------------------------------------------------------------

struct BatchDesc {
    const double *root_mat, *root_rhs; // [h0][pitch], [h0]
    const int32_t *root_pos, *root_var; // [w + h0]
    double *ws_mat, *ws_rhs;           // per node: [hmax][pitch], [hmax]
    int32_t *ws_pos, *ws_var;          // per node: [permmax]
    const int32_t *cut_off, *cut_sign, *cut_var; // cuts of node i: [cut_off[i], cut_off[i+1])
    const double *cut_val;
    int32_t *status, *height;
    double *result;
    long long *pivots;
    int32_t w, n, pitch, h0, hmax, permmax;
    double precision, max_pivots;
};

template <int T>
__global__ __launch_bounds__(T) void batch_kernel(BatchDesc d) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double sh_dyn[]; // colbuf[hmax], prow[pitch]
    double *colbuf = sh_dyn, *prow = sh_dyn + d.hmax;

    const int tid = threadIdx.x, node = blockIdx.x;
    const int w = d.w, n = d.n, pitch = d.pitch, h0 = d.h0;
    const double precision = d.precision, max_pivots = d.max_pivots;
    double *mat = d.ws_mat + (size_t)node * d.hmax * pitch;
    double *rhs = d.ws_rhs + (size_t)node * d.hmax;
    int32_t *pos = d.ws_pos + (size_t)node * d.permmax;
    int32_t *var = d.ws_var + (size_t)node * d.permmax;
    const int c_lo = d.cut_off[node], ncuts = d.cut_off[node + 1] - c_lo;
    const int h = h0 + ncuts;
    const int units = pitch / 2;
    int slot = 0;

    // ---- applyCuts (src/branchAndCut.ts:22-61) ----
    for (int r = 0; r < h0; r++) {
        const double *src = d.root_mat + (size_t)r * pitch;
        double *dst = mat + (size_t)r * pitch;
        for (int u = tid; u < units; u += T)
            *reinterpret_cast<double2 *>(dst + 2 * u) = *reinterpret_cast<const double2 *>(src + 2 * u);
    }
    for (int r = tid; r < h0; r += T) rhs[r] = d.root_rhs[r];
    for (int i = 0; i < ncuts; i++) {
        const double sign = (double)d.cut_sign[c_lo + i], value = d.cut_val[c_lo + i];
        const int p = d.root_pos[d.cut_var[c_lo + i]];
        double *dst = mat + (size_t)(h0 + i) * pitch;
        if (p < w) { // non-basic at the root: sign * x <= sign * value   (:32-35)
            for (int c = tid; c < pitch; c += T) dst[c] = (c == p - 1) ? sign : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * value;
        } else { // basic in root row p - w: substitute that row   (:36-42)
            const double *src = d.root_mat + (size_t)(p - w) * pitch;
            for (int c = tid; c < pitch; c += T) dst[c] = c < n ? -sign * src[c] : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * (value - d.root_rhs[p - w]);
        }
    }
    for (int i = tid; i < w + h; i += T) { // :46-52
        pos[i] = i < w + h0 ? d.root_pos[i] : i;
        var[i] = i < w + h0 ? d.root_var[i] : i;
    }
    __syncthreads();

    // ---- simplex(): src/simplex.ts:106-142 then :66-103 ----
    int phase = 1, status = YALPS_CYCLED;
    double iter = 0.0, result = NAN;
    long long pivots = 0;
    for (;;) {
        if (!(iter < max_pivots)) break; // "cycled"
        int row = 0, col = 0;
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
                status = YALPS_INFEASIBLE;
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
                status = YALPS_OPTIMAL;
                result = round_to_precision(rhs[0], precision);
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
                status = YALPS_UNBOUNDED;
                result = (double)col;
                break;
            }
            row = c.i;
        }
        // ---- pivot(row, col): src/simplex.ts:5-39 ----
        for (int r = tid; r < h; r += T) colbuf[r] = mat[(size_t)r * pitch + col - 1];
        __syncthreads();
        const double q = colbuf[row], rhs_row = rhs[row];
        double *mrow = mat + (size_t)row * pitch;
        for (int c = tid; c < pitch; c += T) {
            const double v = mrow[c];
            const bool nz = fabs(v) > 1e-16;
            const double pn = nz ? v / q : 0.0;
            mrow[c] = (c == col - 1) ? 1.0 / q : pn;
            prow[c] = nz ? pn : __longlong_as_double((long long)FLUSHED);
        }
        __syncthreads(); // (also: everybody has read rhs[row] before it changes)
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
        for (int r = tid; r < h; r += T) {
            if (r == row) {
                rhs[r] = pn_rhs;
            } else if (nz_rhs && fabs(colbuf[r]) > 1e-16) {
                const double prod = colbuf[r] * pn_rhs;
                rhs[r] = rhs[r] - prod;
            }
        }
        for (int u = tid; u < units; u += T) {
            const double2 p = *reinterpret_cast<const double2 *>(prow + 2 * u);
            const bool f0 = (unsigned long long)__double_as_longlong(p.x) != FLUSHED;
            const bool f1 = (unsigned long long)__double_as_longlong(p.y) != FLUSHED;
            const bool has_col = (col - 1) >> 1 == u;
#pragma unroll 4
            for (int r = 0; r < h; r++) {
                const double coef = colbuf[r];
                if (r == row || !(fabs(coef) > 1e-16)) continue; // uniform
                double2 *xp = reinterpret_cast<double2 *>(mat + (size_t)r * pitch + 2 * u);
                double2 x = *xp;
                if (f0) {
                    const double prod = coef * p.x;
                    x.x = x.x - prod;
                }
                if (f1) {
                    const double prod = coef * p.y;
                    x.y = x.y - prod;
                }
                if (has_col) {
                    const double nq = -coef / q;
                    if ((col - 1) & 1)
                        x.y = nq;
                    else
                        x.x = nq;
                }
                *xp = x;
            }
        }
        if (tid == 0) { // :7-12
            const int leaving = var[w + row], entering = var[col];
            var[w + row] = entering;
            var[col] = leaving;
            pos[leaving] = col;
            pos[entering] = w + row;
        }
        iter += 1.0;
        pivots += 1;
        __syncthreads();
    }
    if (tid == 0) {
        d.status[node] = status;
        d.result[node] = result;
        d.pivots[node] = pivots;
        d.height[node] = h;
    }
}
