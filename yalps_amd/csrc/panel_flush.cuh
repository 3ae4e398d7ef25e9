// panel_flush.cuh -- the sweep of the delayed-update kernels (stream3_kernel, dshard_kernel): every touched row of a
// workgroup streamed once, all pending eliminations applied in registers, the pending normalised pivot rows staged in LDS
// one column PANEL at a time.
// Part of libyalps_hip.so; included by persistent_stream3.hip / persistent_dshard.hip inside their unnamed namespaces.
#pragma once

// Round 2 read the pending rows from L2 for every two to four (half-)rows in flight: with 8 pending pivots that is 2-4 x the
// rows' own traffic through the L2s, and eight dependent L2 round trips per batch -- 16385^2 swept at 4.6 TB/s where the
// bare in-place sweep reaches 6.1 (profiles/r03_stream3_stages_16384.json: the sweep is 82 % of a pivot there).  Here a
// panel of PU 16-byte units (2 PU columns) of ALL pending rows is copied into LDS once per workgroup, and the workgroup's
// rows pass through registers one panel-wide segment at a time: L2 traffic for the pending rows drops to npend x pitch per
// workgroup and sweep, a pending entry costs one ds_read_b128 per D rows, and a lane holds one unit per row in flight, so
// the sweep no longer decides the kernel's register budget (tools/micro/panel_sweep.hip, profiles/r03_panel_sweep.txt:
// D = 4 rows in flight, panels as wide as LDS allows; 16 pending pivots of 1024 columns = 128 KB).
//
// The arithmetic per element is the reference's, pending pivot by pending pivot, oldest first (src/simplex.ts:14-38):
//   the row that was pending pivot p's pivot row:  x = p-th normalised row (0.0 where pivot() flushed, :17-24)
//   a row with |coef| > 1e-16 (:31):               x = x - coef * pn, product and difference rounded separately (:33), only where
//                                                  the pivot row's entry was not flushed (nonZeroColumns, :18-23)
//   its entry of the pivot column:                 1 / quotient resp. -coef / quotient (:25, :36), computed when the pivot was decided
// `panel` is LDS for MAXD x 2 PU doubles; colv / nqv are [npend][rpw] LDS arrays of my rows' pivot-column entries as they
// were and what replaces them; pl[p] = my slot of pending pivot p's pivot row or -1, pc[p] = its pivot column (mat index);
// tlist[0 .. nt) = my row slots touched by at least one pending pivot.  Rows and pending rows are addressed through buffer
// descriptors of one row (rsrc_of): units past the pitch read as 0.0 and their stores are dropped.
template <int T, int PU, int D, bool NT, typename RsrcOf>
__device__ __forceinline__ void panel_flush(double *mat, int pitch, int b, int NB, const double *pend0, int npend, const double *colv,
                                            const double *nqv, int rpw, const int *pl, const int *pc, const int *tlist, int nt, double *panel,
                                            RsrcOf rsrc_of) {
    static_assert(PU % 64 == 0 && (PU >= T ? PU % T == 0 : T % PU == 0), "a wave stays within one row segment");
    constexpr int RS = PU < T ? T / PU : 1; // rows side by side
    constexpr int U = PU > T ? PU / T : 1;  // units per lane and row
    constexpr int LU = PU < T ? PU : T;     // lanes across a row segment
    constexpr int AUX = NT ? AUX_NT : AUX_PLAIN;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid)); // (opaque: the lane's LDS and row offsets are recomputed here, not hoisted out of the caller's pivot loop and kept -- or spilled -- there)
    const int sub = tid / LU, lane = tid % LU;
    const int units = pitch >> 1, npanel = (units + PU - 1) / PU;
    for (int pnl = 0; pnl < npanel; pnl++) {
        const int u0 = pnl * PU;
        __syncthreads(); // (everybody is through with the previous panel -- and, the first time, with whatever used this LDS before)
        for (int i = tid; i < npend * PU; i += T) { // (a wave's 64 units belong to one pending row: PU is a multiple of 64)
            const int p = i / PU, u = i - p * PU;
            const double2 v = row_ld16<AUX_PLAIN>(rsrc_of(pend0 + (size_t)p * pitch), 16 * (u0 + u), 0);
            *reinterpret_cast<double2 *>(panel + (size_t)p * 2 * PU + 2 * u) = v;
        }
        __syncthreads();
        // per wave and pending pivot: nothing of my lanes' units was flushed -> the select-free path (:31's inner loop as two
        // fp64 instructions per element); the pivot-column patch of this panel, if the column lies in it
        unsigned fastmask = 0;
        for (int p = 0; p < npend; p++) {
            bool fl = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double2 pn = *reinterpret_cast<const double2 *>(panel + (size_t)p * 2 * PU + 2 * (lane + u * LU));
                fl = fl || (unsigned long long)__double_as_longlong(pn.x) == FLUSHED || (unsigned long long)__double_as_longlong(pn.y) == FLUSHED;
            }
            if (__builtin_amdgcn_ballot_w64(fl) == 0) fastmask |= 1u << p;
        }
        for (int k0 = 0; k0 < nt; k0 += RS * D) {
            double2 x[D][U];
            int ri[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int k = k0 + d * RS + sub;
                ri[d] = k < nt ? tlist[k] : -1;
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * (ri[d] < 0 ? tlist[0] : ri[d])) * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) x[d][u] = row_ld16<AUX>(rs, 16 * (u0 + lane + u * LU), 0);
            }
#pragma unroll 1
            for (int p = 0; p < npend; p++) {
                double2 pn[U];
#pragma unroll
                for (int u = 0; u < U; u++) pn[u] = *reinterpret_cast<const double2 *>(panel + (size_t)p * 2 * PU + 2 * (lane + u * LU));
                const int colxp = pc[p], lslotp = pl[p];
                const bool fastp = (fastmask >> p) & 1u;
                const int pcu = (colxp >> 1) - u0; // the pivot column's unit within this panel (uniform; in range or not)
                const bool col_here = (unsigned)pcu < (unsigned)PU;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    if (ri[d] < 0) continue; // (uniform per wave)
                    const double coef = colv[p * rpw + ri[d]];
                    const bool piv = ri[d] == lslotp;
                    if (!(piv || fabs(coef) > 1e-16)) continue; // :31
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        double2 &xv = x[d][u];
                        if (fastp && !piv) {
                            const double px = coef * pn[u].x, py = coef * pn[u].y;
                            xv.x = xv.x - px;
                            xv.y = xv.y - py;
                        } else {
                            const bool f0 = (unsigned long long)__double_as_longlong(pn[u].x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pn[u].y) != FLUSHED;
                            if (piv) {
                                xv.x = f0 ? pn[u].x : 0.0;
                                xv.y = f1 ? pn[u].y : 0.0;
                            } else {
                                const double px = coef * pn[u].x, py = coef * pn[u].y;
                                const double nx = xv.x - px, ny = xv.y - py;
                                xv.x = f0 ? nx : xv.x;
                                xv.y = f1 ? ny : xv.y;
                            }
                        }
                    }
                    if (col_here) { // (uniform) :25, :36 -- the one element of the row that the pivot column replaces
                        const double patch = nqv[p * rpw + ri[d]];
#pragma unroll
                        for (int u = 0; u < U; u++)
                            if (pcu == lane + u * LU) {
                                if (colxp & 1)
                                    x[d][u].y = patch;
                                else
                                    x[d][u].x = patch;
                            }
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                if (ri[d] < 0) continue;
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * ri[d]) * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) row_st16<AUX>(rs, 16 * (u0 + lane + u * LU), 0, x[d][u]);
            }
        }
    }
    __syncthreads(); // (the panel LDS may be reused by the caller)
}
