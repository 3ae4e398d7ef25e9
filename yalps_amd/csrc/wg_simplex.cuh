// wg_simplex.cuh -- the whole two-phase simplex loop run by ONE workgroup on a tableau it owns
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// wg_simplex: simplex() (src/simplex.ts:106-142, then :66-103) + pivot() (:5-39) on a tableau that
// only this workgroup touches -- a branch-and-cut node in its HBM workspace (batch_kernel), or a
// small tableau held entirely in LDS (small_kernel / batch_kernel's LDS variant).  Only
// workgroup-local synchronisation.  Layout: element (r, c >= 1) at mat[r * lp + c - 1], column 0
// at rhs[r]; columns [n, pcols) of every row are zero padding (pcols even, <= lp).
// Per pivot: the pivot column is gathered into colbuf first (so rows can then be updated in
// place), the pivot row is normalised into prow (FLUSHED marks entries pivot() zeroed), and the
// sweep maps lanes to (16-byte column unit, row group): U lanes across the units, T / U row groups.
// ------------------------------------------------------------------------------------------
constexpr int WG_HISTORY_FULL = -100; // checkCycles: the pivot history buffer is too short (host grows it and reruns)

struct WgResult {
    int status;
    double result;
    long long pivots;
};

// CHECK: options.checkCycles -- hasCycle (src/simplex.ts:44-63) before every pivot; the history of
// (leaving, entering) variables of the current phase is kept in hist_l / hist_e (hist_cap entries,
// global memory), every lane tests a share of the candidate cycle lengths.
template <int T, bool CHECK>
__device__ __attribute__((always_inline)) WgResult wg_simplex(double *mat, double *rhs, int32_t *pos, int32_t *var,
                                                              double *colbuf, double *prow, double (*sk)[16],
                                                              int (*si)[16], int w, int n, int lp, int pcols, int h,
                                                              int U, double precision, double max_pivots,
                                                              int32_t *hist_l = nullptr, int32_t *hist_e = nullptr,
                                                              long long hist_cap = 0) {
    const int tid = threadIdx.x;
    const int units = pcols / 2;
    const int u0 = tid % U, g0 = tid / U, G = T / U; // U is a power of two <= T
    int slot = 0;
    int phase = 1;
    WgResult out = {YALPS_CYCLED, NAN, 0};
    double iter = 0.0;
    long long hist_len = 0; // pivots of the current phase (src/simplex.ts:67,107: one history per phase)
    for (;;) {
        if (!(iter < max_pivots)) break; // "cycled" (:102,141)
        int row = 0, col = 0;
        if (phase == 1) {
            KI c = {INFINITY, INT_MAX}; // :111-119
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
                phase = 2; // :120
                iter = 0.0;
                hist_len = 0;
                continue;
            }
            row = c.i;
            const double *mrow = mat + (size_t)row * lp;
            KI e = {INFINITY, INT_MAX}; // :123-134
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
                out.status = YALPS_INFEASIBLE; // :135
                break;
            }
            col = e.i;
        } else {
            KI pr = {INFINITY, INT_MAX}; // :71-79
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
                out.status = YALPS_OPTIMAL; // :80
                out.result = round_to_precision(rhs[0], precision);
                break;
            }
            col = pr.i;
            KI c = {INFINITY, INT_MAX}; // :83-95, closed form of the early break
            for (int r = 1 + tid; r < h; r += T) {
                const double value = mat[(size_t)r * lp + col - 1];
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
                out.status = YALPS_UNBOUNDED; // :96
                out.result = (double)col;
                break;
            }
            row = c.i;
        }
        if (CHECK) { // :98,137
            if (hist_len >= hist_cap) {
                out.status = WG_HISTORY_FULL;
                break;
            }
            if (tid == 0) {
                hist_l[hist_len] = var[w + row];
                hist_e[hist_len] = var[col];
            }
            __syncthreads();
            const long long len = hist_len + 1;
            bool found = false;
            for (long long length = 6 + tid; length <= len / 2 && !found; length += T) {
                bool cycle = true;
                for (long long i = 0; i < length; i++) {
                    const long long item = len - 1 - i;
                    if (hist_l[item] != hist_l[item - length] || hist_e[item] != hist_e[item - length]) {
                        cycle = false;
                        break;
                    }
                }
                found = cycle;
            }
            hist_len = len;
            if (__syncthreads_or(found ? 1 : 0)) break; // "cycled", NaN
        }
        // ---- pivot(row, col): src/simplex.ts:5-39 ----
        for (int r = tid; r < h; r += T) colbuf[r] = mat[(size_t)r * lp + col - 1];
        __syncthreads();
        const double q = colbuf[row], rhs_row = rhs[row];
        double *mrow = mat + (size_t)row * lp;
        for (int c = tid; c < pcols; c += T) {
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
        for (int u = u0; u < units; u += U) {
            const double2 p = *reinterpret_cast<const double2 *>(prow + 2 * u);
            const bool f0 = (unsigned long long)__double_as_longlong(p.x) != FLUSHED;
            const bool f1 = (unsigned long long)__double_as_longlong(p.y) != FLUSHED;
            const bool has_col = (col - 1) >> 1 == u;
#pragma unroll 4
            for (int r = g0; r < h; r += G) {
                const double coef = colbuf[r];
                if (r == row || !(fabs(coef) > 1e-16)) continue;
                double2 *xp = reinterpret_cast<double2 *>(mat + (size_t)r * lp + 2 * u);
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
        out.pivots += 1;
        __syncthreads();
    }
    return out;
}

// lanes across the 16-byte column units of a row: the power of two >= units, capped at T
__host__ __device__ inline int wg_unit_lanes(int units, int T) {
    int U = 1;
    while (U < units && U < T) U *= 2;
    return U;
}

// ------------------------------------------------------------------------------------------
// small_kernel: a tableau that fits in the LDS of one CU is solved by ONE workgroup from start to
// finish -- no cross-workgroup hand-off per pivot (the resident kernel's L2 exchange costs ~5 us a
// pivot whatever the size), one launch per solve.  Input and output are described by strides so
// that the same kernel serves the HBM layout (mat[pitch] + rhs) and the reference's host layout
// (row-major width*height, column 0 = RHS) read and written IN PLACE in pinned host memory over
// PCIe -- the drop-in call then needs no copies, one launch and one synchronisation.
// ------------------------------------------------------------------------------------------
struct SmallResult {
    int32_t status, pad_;
    double result;
    long long pivots;
};

struct SmallDesc {
    double *mat, *rhs;            // element (r, c >= 1) at mat[r * pitch + c - 1]; column 0 at rhs[r * rhs_stride]
    long long pitch, rhs_stride;
    int32_t *pos, *var;           // [w + h]
    SmallResult *res;
    int32_t w, n, h, lp;          // lp: LDS row pitch in doubles (even, >= pcols)
    double precision, max_pivots;
    int32_t *hist_l, *hist_e;     // checkCycles: pivot history, hist_cap entries each
    long long hist_cap;
};

constexpr size_t SMALL_LDS_MAX = 150 * 1024; // of the 160 KB of LDS per CU
__host__ __device__ inline int small_pcols(int n) { return (n + 1) & ~1; }
// an odd number of 16-byte units per row: walking a column then touches distinct LDS banks
__host__ __device__ inline int small_lds_pitch(int n) { return small_pcols(n) | 2; }
__host__ __device__ inline size_t small_lds_bytes(int w, int h) {
    const int lp = small_lds_pitch(w - 1);
    return sizeof(double) * ((size_t)h * lp + 2 * (size_t)h + (size_t)lp) + sizeof(int32_t) * 2 * ((size_t)w + h + 1);
}

template <int T, bool CHECK>
__global__ __launch_bounds__(T) void small_kernel(SmallDesc d) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double sh_dyn[];
    const int tid = threadIdx.x, w = d.w, n = d.n, h = d.h, lp = d.lp, pcols = small_pcols(n);
    double *mat = sh_dyn, *rhs = mat + (size_t)h * lp, *colbuf = rhs + h, *prow = colbuf + h;
    int32_t *pos = reinterpret_cast<int32_t *>(prow + lp), *var = pos + ((w + h + 1) & ~1);

    const int Uc = wg_unit_lanes(pcols, T), cu0 = tid % Uc, cg0 = tid / Uc, CG = T / Uc;
    for (int r = cg0; r < h; r += CG) {
        const double *src = d.mat + (size_t)r * d.pitch;
        for (int c = cu0; c < pcols; c += Uc) mat[(size_t)r * lp + c] = c < n ? src[c] : 0.0;
    }
    for (int r = tid; r < h; r += T) rhs[r] = d.rhs[(size_t)r * d.rhs_stride];
    for (int i = tid; i < w + h; i += T) {
        pos[i] = d.pos[i];
        var[i] = d.var[i];
    }
    __syncthreads();

    const WgResult out = wg_simplex<T, CHECK>(mat, rhs, pos, var, colbuf, prow, sk, si, w, n, lp, pcols, h,
                                              wg_unit_lanes(pcols / 2, T), d.precision, d.max_pivots, d.hist_l, d.hist_e,
                                              d.hist_cap);
    __syncthreads();
    if (CHECK && out.status == WG_HISTORY_FULL) { // leave the caller's tableau untouched: the host reruns
        if (tid == 0) d.res->status = WG_HISTORY_FULL;
        return;
    }
    for (int r = cg0; r < h; r += CG) {
        double *dst = d.mat + (size_t)r * d.pitch;
        for (int c = cu0; c < n; c += Uc) dst[c] = mat[(size_t)r * lp + c];
    }
    for (int r = tid; r < h; r += T) d.rhs[(size_t)r * d.rhs_stride] = rhs[r];
    for (int i = tid; i < w + h; i += T) {
        d.pos[i] = pos[i];
        d.var[i] = var[i];
    }
    if (tid == 0) {
        d.res->status = out.status;
        d.res->result = out.result;
        d.res->pivots = out.pivots;
    }
}
