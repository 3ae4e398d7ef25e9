// dsweep_kernel.cuh -- the sweep of a row shard with delayed row updates (dshard_kernel.cuh) as a LAUNCH OF ITS OWN, with the
// rows mapped to workgroups the other way round.
// Part of libyalps_hip.so; included by persistent_dsweep.hip inside its unnamed namespace.
#pragma once

// Inside dshard_kernel a workgroup sweeps ITS rows (b, b + NB, ...: the rows whose scalars it holds in LDS) panel by panel: per
// panel a fill of npend x 8 KB and two barriers, whatever the number of rows behind them -- with 8 rows per workgroup (a rank's
// share of eight at 16385 columns) the panels cost more than they save, and with 64 the barriers keep the waves of a workgroup
// in step: all load, all compute, all store, and a sweep costs the SUM of its memory time and its arithmetic.
// The shard's launches end at every pivot anyway, and everything a sweep needs is in global memory between them (the
// pending rows d.dpend, my rows' pivot-column entries and what replaces them d.dcolv / d.dnqv, the pending pivots' rows and
// columns in DelayState), so the sweep does not have to keep the step kernel's map:
//   workgroup g = (panel g % npan, row block g / npan): ONE fill of its panel of the pending rows for its whole life, no barrier
//   afterwards; its waves take pairs of rows of the block from a counter, each at its own pace;
//   per pair of rows: which pending pivots touch them (32 lanes read the 2 x 16 coefficients, one ballot), rows nobody touches are
//   neither read nor written; the coefficients stay in those lanes and reach the arithmetic through v_readlane;
//   the arithmetic is panel_flush.cuh's, element for element (src/simplex.ts:14-38): the usual pending pivot as straight-line code.
// The step kernel that finds `depth` pivots pending leaves them pending (Desc::ext_sweep); the host enqueues this kernel behind
// it; the workgroup that finishes last marks nothing pending.  Should the launch be missing, the next step kernel sweeps itself.
template <bool NT>
__global__ __launch_bounds__(512) void dshard_sweep_kernel(Desc d, int parity) {
    constexpr int T = 512, PU = DSHARD_PANEL_UNITS, LU = 64, U = PU / LU, UH = U / 2, D = 2, MAXD = DSHARD_MAXD, CH = 8;
    constexpr int AUX = NT ? AUX_NT : AUX_PLAIN;
    static_assert(U == 8 && MAXD <= 16, "eight units per lane and row; the masks of two rows in one ballot");
    extern __shared__ __attribute__((aligned(16))) double panel[]; // [npend][2 PU]
    __shared__ unsigned sh_slow[T / 64];
    __shared__ int sh_next, sh_pl[MAXD], sh_pc[MAXD];
    const int tid = threadIdx.x, lane = tid & 63;
    const YState *S = d.st + (parity ^ 1); // what the step kernel of this pivot left
    DelayState *DS = d.dstate + (parity ^ 1);
    const int depth = d.delay_depth < 1 ? 1 : d.delay_depth > MAXD ? MAXD : d.delay_depth;
    const int npend = DS->npend;
    if (S->status != RUNNING || npend != depth) return; // (the same for every workgroup: nobody changes it before the last one is through)
    const int h = d.cst->height, pitch = d.pitch, hcap = d.hcap, units = pitch >> 1, row_bytes = pitch * 8;
    double *mat = d.mat[S->mbuf];
    const int npan = (units + PU - 1) / PU, nrb = (int)gridDim.x / npan; // (the host launches npan x nrb workgroups, nrb >= 1)
    const int g = blockIdx.x, pnl = g % npan, rb = g / npan, u0 = pnl * PU;
    const int rpb = (h - 1 + nrb - 1) / nrb, r_lo = 1 + rb * rpb, r_hi = r_lo + rpb < h ? r_lo + rpb : h; // local rows [r_lo, r_hi) of mine; row 0 is the objective row
    auto rsrc_of = [&](const double *row_ptr) __attribute__((always_inline)) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(row_ptr);
        const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(u), 0, row_bytes, 0x00020000);
    };
    if (g < npan * nrb) { // (uniform; a grid that does not divide leaves its last workgroups without work)
        // ---- the panel: all pending rows' units [u0, u0 + PU), once; which of them hold a flushed entry (:17-24) ----
        if (tid < MAXD) {
            sh_pl[tid] = tid < npend ? DS->pl[tid] : -1; // local row of the pivot row (-1: another rank's)
            sh_pc[tid] = tid < npend ? DS->pc[tid] : 0;
        }
        if (tid == 0) sh_next = 0;
        {
            const int items = npend * PU;
            const unsigned long long pa0 = reinterpret_cast<unsigned long long>(d.dpend);
            const unsigned long long pau = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pa0);
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(pau), 0, __builtin_amdgcn_readfirstlane(npend * row_bytes), 0x00020000);
            unsigned slow = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < items; i0 += CH * T) {
                double2 v[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = i0 + k * T + tid;
                    const int pp = __builtin_amdgcn_readfirstlane(i / PU);
                    const int un = u0 + i - pp * PU; // (past the pitch: 0.0, as a row's own descriptor returns it)
                    if (i < items) v[k] = un < units ? row_ld16<AUX_PLAIN>(rs0, pp * row_bytes + 16 * un, 0) : double2{0.0, 0.0};
                }
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = i0 + k * T + tid;
                    const int pp = __builtin_amdgcn_readfirstlane(i / PU);
                    if (i < items) {
                        *reinterpret_cast<double2 *>(panel + (size_t)pp * 2 * PU + 2 * (i - pp * PU)) = v[k];
                        const bool fl = (unsigned long long)__double_as_longlong(v[k].x) == FLUSHED || (unsigned long long)__double_as_longlong(v[k].y) == FLUSHED;
                        if (__builtin_amdgcn_ballot_w64(fl) != 0) slow |= 1u << pp;
                    }
                }
            }
            if (lane == 0) sh_slow[tid >> 6] = slow;
        }
        // the objective row is not swept: the replica the step kernels keep IS that row with every pending pivot applied
        if (rb == 0) {
            const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(d.obj[S->pbuf]), rs_0 = rsrc_of(mat);
            const double2 o = row_ld16<AUX_PLAIN>(rs_o, 16 * (u0 + tid), 0);
            row_st16<AUX>(rs_0, 16 * (u0 + tid), 0, o);
        }
        __syncthreads();
        unsigned fastmask, colmask = 0;
        {
            unsigned slow = 0;
#pragma unroll
            for (int w = 0; w < T / 64; w++) slow |= sh_slow[w];
            fastmask = ~slow;
        }
        for (int p = 0; p < npend; p++) {
            const int pcu = (__builtin_amdgcn_readfirstlane(sh_pc[p]) >> 1) - u0;
            if ((unsigned)pcu < (unsigned)PU) colmask |= 1u << p;
        }
        const double *pan = panel + 2 * lane;
        // ---- my block's rows, two at a time per wave, no barrier from here on ----
        // Two register sets: the rows of the NEXT pair are on their way while the pending pivots are applied to this one (the kernel
        // has the registers for it, which the step kernel -- with its own state live across the sweep -- has not).
        struct Pair {
            int r0;            // first local row of the pair (>= r_hi: the block is done)
            unsigned act[D], piv[D]; // pending pivots that touch the row (:31, or its pivot row) / that have it as their pivot row
            bool ok[D];        // a row nobody touches -- or past the block -- is neither read nor written
            int rr[D], cq_lo, cq_hi; // lanes 0 .. 15: row r0's entries of the pending pivots' pivot columns as they were, 16 .. 31: row r0 + 1's
        };
        auto fetch = [&](Pair &P, double2 (&x)[D][U]) __attribute__((always_inline)) {
            int k0 = 0;
            if (lane == 0) k0 = __hip_atomic_fetch_add(&sh_next, D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            P.r0 = r_lo + __builtin_amdgcn_readlane(k0, 0);
            if (P.r0 >= r_hi) return; // (uniform)
            const int q = lane & 15, dsel = (lane >> 4) & 1, rq = P.r0 + dsel;
            const bool mine = lane < 32 && q < npend && rq < r_hi;
            double cq = 0.0;
            if (mine) cq = d.dcolv[(size_t)q * hcap + rq];
            const bool pvq = mine && sh_pl[q] == rq;
            const unsigned long long mt = __builtin_amdgcn_ballot_w64(mine && (pvq || fabs(cq) > 1e-16)), mp = __builtin_amdgcn_ballot_w64(pvq);
            P.cq_lo = __double2loint(cq);
            P.cq_hi = __double2hiint(cq);
#pragma unroll
            for (int dd = 0; dd < D; dd++) {
                P.act[dd] = (unsigned)(mt >> (16 * dd)) & 0xffffu;
                P.piv[dd] = (unsigned)(mp >> (16 * dd)) & 0xffffu;
                P.ok[dd] = P.act[dd] != 0;
            }
            if (!P.ok[0] && !P.ok[1]) return; // (uniform)
#pragma unroll
            for (int dd = 0; dd < D; dd++) {
                P.rr[dd] = P.ok[dd] ? P.r0 + dd : (P.ok[0] ? P.r0 : P.r0 + 1); // (the row that is not there: the other one's addresses, its registers never stored)
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)P.rr[dd] * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) x[dd][u] = row_ld16<AUX>(rs, 16 * (u0 + lane + u * LU), 0);
            }
        };
        auto process = [&](const Pair &P, double2 (&x)[D][U]) __attribute__((always_inline)) {
            if (!P.ok[0] && !P.ok[1]) return; // (uniform)
            unsigned plain = fastmask & ~colmask;
#pragma unroll
            for (int dd = 0; dd < D; dd++)
                if (P.ok[dd]) plain &= P.act[dd] & ~P.piv[dd];
            auto rd_units = [&](int p, int ub, double2 (&pn)[UH]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < UH; u++) pn[u] = *reinterpret_cast<const double2 *>(pan + (size_t)p * 2 * PU + 2 * (ub + u) * LU);
            };
            auto coef_of = [&](int p, int dd) __attribute__((always_inline)) { // (uniform lane index: v_readlane)
                return __hiloint2double(__builtin_amdgcn_readlane(P.cq_hi, p + 16 * dd), __builtin_amdgcn_readlane(P.cq_lo, p + 16 * dd));
            };
            auto straight = [&](const double (&cf_c)[D], int ub, const double2 (&pn_c)[UH]) __attribute__((always_inline)) {
#pragma unroll
                for (int dd = 0; dd < D; dd++)
#pragma unroll
                    for (int u = 0; u < UH; u++) {
                        double2 &xv = x[dd][ub + u];
                        const double px = cf_c[dd] * pn_c[u].x, py = cf_c[dd] * pn_c[u].y;
                        xv.x = xv.x - px;
                        xv.y = xv.y - py;
                    }
            };
            auto work = [&](int p, const double (&cf_c)[D], int ub, const double2 (&pn_c)[UH]) __attribute__((always_inline)) {
                const int colxp = sh_pc[p];
                const bool fastp = (fastmask >> p) & 1u;
                const int pcu = (colxp >> 1) - u0; // the pivot column's unit within this panel (uniform; in range or not)
                const bool col_here = (unsigned)pcu < (unsigned)PU;
#pragma unroll
                for (int dd = 0; dd < D; dd++) {
                    if (!P.ok[dd] || !((P.act[dd] >> p) & 1u)) continue; // (uniform) :31
                    const double coef = cf_c[dd];
                    const bool pv = (P.piv[dd] >> p) & 1u;
                    if (fastp && !pv) { // nothing of this panel's slice of the pivot row was flushed: two instructions per element
#pragma unroll
                        for (int u = 0; u < UH; u++) {
                            double2 &xv = x[dd][ub + u];
                            const double px = coef * pn_c[u].x, py = coef * pn_c[u].y;
                            xv.x = xv.x - px;
                            xv.y = xv.y - py;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < UH; u++) {
                            double2 &xv = x[dd][ub + u];
                            const double2 pn = pn_c[u];
                            const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
                            if (pv) {
                                xv.x = f0 ? pn.x : 0.0;
                                xv.y = f1 ? pn.y : 0.0;
                            } else {
                                const double px = coef * pn.x, py = coef * pn.y;
                                const double nx = xv.x - px, ny = xv.y - py;
                                xv.x = f0 ? nx : xv.x;
                                xv.y = f1 ? ny : xv.y;
                            }
                        }
                    }
                    if (col_here) { // (one panel in npan) :25, :36 -- the one element of the row that the pivot column replaces
                        const double patch = d.dnqv[(size_t)p * hcap + P.rr[dd]];
#pragma unroll
                        for (int u = 0; u < UH; u++)
                            if (pcu == lane + (ub + u) * LU) {
                                if (colxp & 1)
                                    x[dd][ub + u].y = patch;
                                else
                                    x[dd][ub + u].x = patch;
                            }
                    }
                }
            };
            // (the LDS reads run half a pending pivot ahead of the arithmetic, as in panel_flush.cuh)
            double2 pa[UH], pb[UH];
            rd_units(0, 0, pa);
#pragma unroll 1
            for (int p = 0; p < npend; p++) {
                const int pnx = p + 1 < npend ? p + 1 : p;
                const bool is_plain = (plain >> p) & 1u;
                double cf[D];
#pragma unroll
                for (int dd = 0; dd < D; dd++) cf[dd] = coef_of(p, dd);
                rd_units(p, UH, pb);
                __builtin_amdgcn_sched_barrier(0);
                if (is_plain)
                    straight(cf, 0, pa);
                else
                    work(p, cf, 0, pa);
                __builtin_amdgcn_sched_barrier(0);
                rd_units(pnx, 0, pa);
                __builtin_amdgcn_sched_barrier(0);
                if (is_plain)
                    straight(cf, UH, pb);
                else
                    work(p, cf, UH, pb);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int dd = 0; dd < D; dd++) {
                if (!P.ok[dd]) continue;
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)P.rr[dd] * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) row_st16<AUX>(rs, 16 * (u0 + lane + u * LU), 0, x[dd][u]);
            }
        };
        Pair A, B;
        A.ok[0] = A.ok[1] = B.ok[0] = B.ok[1] = false;
        double2 xa[D][U], xb[D][U];
        fetch(A, xa);
#pragma unroll 1
        while (A.r0 < r_hi) {
            B.ok[0] = B.ok[1] = false;
            fetch(B, xb);
            process(A, xa);
            if (B.r0 >= r_hi) break;
            A.ok[0] = A.ok[1] = false;
            fetch(A, xa);
            process(B, xb);
        }
    }
    // ---- the workgroup that is through last marks nothing pending (the next step kernel reads it behind the launch boundary) ----
    __syncthreads();
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(&DS->pad_[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&DS->pad_[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&DS->npend, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
