// panel_flush.cuh -- the sweep of the delayed-update kernels (stream3_kernel, dshard_kernel): every touched row of a
// workgroup streamed once, all pending eliminations applied in registers, the pending normalised pivot rows staged in LDS
// one column PANEL at a time.
// Part of libyalps_hip.so; included by persistent_stream3.hip / persistent_dshard.hip inside their unnamed namespaces.
#pragma once
#ifdef YALPS_STAMPS // (diagnostic build: the callers' stage sums continue inside the sweep, stages 11-18; tools/delayed_stages.py)
#define YSTAMP_PARAMS , unsigned long long (&st_acc)[20], unsigned long long &st_last
#define YSTAMP_ARGS , st_acc, st_last
#else
#define YSTAMP_PARAMS
#define YSTAMP_ARGS
#endif

// Round 2 read the pending rows from L2 for every two to four (half-)rows in flight: with 8 pending pivots that is 2-4 x the
// rows' own traffic through the L2s, and eight dependent L2 round trips per batch -- 16385^2 swept at 4.6 TB/s where the
// bare in-place sweep reaches 6.1 (profiles/r03_stream3_stages_16384.json: the sweep is 82 % of a pivot there).  Here a
// panel of PU 16-byte units (2 PU columns) of ALL pending rows is copied into LDS once per workgroup, and the workgroup's
// rows pass through registers one panel-wide segment at a time: L2 traffic for the pending rows drops to npend x pitch per
// workgroup and sweep, a pending entry costs one ds_read_b128 per D rows, and a lane holds one unit per row in flight, so
// the sweep no longer decides the kernel's register budget (tools/micro/panel_sweep.hip, profiles/r03_panel_sweep.txt:
// panels as wide as LDS allows; 16 pending pivots of 1024 columns = 128 KB; D rows in flight per lane).
// What it took to make that fast, and what was tried and dropped: DESIGN.md 4.9b / 4.9c; the inner loop alone: tools/micro/lds_axpy.hip.
//
// The arithmetic per element is the reference's, pending pivot by pending pivot, oldest first (src/simplex.ts:14-38):
//   the row that was pending pivot p's pivot row:  x = p-th normalised row (0.0 where pivot() flushed, :17-24)
//   a row with |coef| > 1e-16 (:31):               x = x - coef * pn, product and difference rounded separately (:33), only where
//                                                  the pivot row's entry was not flushed (nonZeroColumns, :18-23)
//   its entry of the pivot column:                 1 / quotient resp. -coef / quotient (:25, :36), computed when the pivot was decided
// `panel` is LDS for MAXD x 2 PU doubles; colv / nqv are [npend][rpw] LDS arrays of my rows' pivot-column entries as they
// were and what replaces them; pl[p] = my slot of pending pivot p's pivot row or -1, pc[p] = its pivot column (mat index);
// tlist[0 .. nt) = my row slots touched by at least one pending pivot, tmask[k] = for row tlist[k], bit p: pending pivot p touches it
// (:31 or its pivot row); tpiv[k], bit p: it is p's pivot row.  Rows and pending rows are addressed through buffer
// descriptors of one row (rsrc_of): units past the pitch read as 0.0 and their stores are dropped.
template <int T, int PU, int LU, int D, int SETS, bool NT, int CH = 8, bool DYN = true, typename RsrcOf>
__device__ __forceinline__ void panel_flush(double *mat, int pitch, int b, int NB, const double *pend0, int npend, const double *colv,
                                            const double *nqv, int rpw, const int *pl, const int *pc, const int *tlist, const int *tmask, const int *tpiv, int nt, double *panel,
                                            RsrcOf rsrc_of YSTAMP_PARAMS) {
    // LU lanes across a row segment of PU units: U = PU / LU units per lane and row, RS = T / LU rows side by side.  One WAVE per
    // row (LU = 64, eight units per lane) is what keeps the sweep off the vector ALU: what a pending pivot costs a row apart
    // from its elements -- coefficient, skip test (:31), pivot-row and pivot-column tests, branches: ~25 instructions -- is
    // then spread over 16 elements at 2 instructions each; with all 512 lanes across one row (one unit per lane) it was spread
    // over two, the select-free path was if-converted away, and the sweep ran VALU-bound at 2.7 TB/s.
    static_assert(LU % 64 == 0 && PU % LU == 0 && T % LU == 0 && PU % 64 == 0, "a wave stays within one row segment");
    constexpr int RS = T / LU; // rows side by side
    constexpr int U = PU / LU; // units per lane and row
    constexpr int AUX = NT ? AUX_NT : AUX_PLAIN;
    __shared__ unsigned sh_slow[T / 64]; // per wave of the fill: pending rows with a flushed entry among the units it copied
    __shared__ int sh_next;              // the next pair of rows of this panel nobody has taken yet (the waves take them as they get free)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid)); // (opaque: the lane's LDS and row offsets are recomputed here, not hoisted out of the caller's pivot loop and kept -- or spilled -- there)
    const int sub = tid / LU, lane = tid % LU;
    // A row of 2^k + 1 columns (every BASELINE configuration) is k' full panels and ONE more unit: a whole panel pass -- fill, two
    // barriers, every wave's trips through the pending pivots -- for 16 bytes of every row.  Up to TAIL units behind the last full
    // panel go through the loop behind the panels instead: a lane per (row, unit), the pending rows' units straight from L2.
    constexpr int TAIL = 64;
    const int units = pitch >> 1, rem = units % PU, ntail = (units > PU && rem <= TAIL) ? rem : 0;
    const int npanel = (units - ntail + PU - 1) / PU;
    for (int pnl = 0; pnl < npanel; pnl++) {
        const int u0 = pnl * PU;
        __syncthreads(); // (everybody is through with the previous panel -- and, the first time, with whatever used this LDS before)
        YSTAMP(18); // sweep: barrier in front of the fill (the other waves' trips)
        // The fill: CH loads of a lane in flight behind one wait (a load, its wait, its LDS store, per pending row, was a chain of
        // npend L2 round trips per panel -- with the test for flushed entries below, a chain of LDS round trips per unit, it
        // made a pending pivot cost a sweep of 4097^2 5.4 us where its arithmetic is 0.9).  One descriptor over all pending rows
        // (a row per load through the instruction's scalar offset does not work: gfx950 range-checks it against the row's length).
        // Per wave and pending row: was any of the 64 units this wave copied flushed (src/simplex.ts:17-24)?  OR-ed over the
        // waves behind the barrier: nothing of this panel's slice of a pending pivot row was flushed -> the select-free path
        // (:31's inner loop as two fp64 instructions per element).
        {
            // (CH: pending rows' loads of a lane in flight at once)
            const int items = npend * PU, row_bytes = pitch * 8;
            const unsigned long long pa0 = reinterpret_cast<unsigned long long>(pend0); // (uniform: into scalar registers)
            const unsigned long long pau = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pa0);
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(pau), 0, __builtin_amdgcn_readfirstlane(npend * row_bytes), 0x00020000);
            unsigned slow = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < items; i0 += CH * T) {
                double2 v[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = i0 + k * T + tid; // (a wave's 64 units belong to one pending row: PU is a multiple of 64)
                    const int pp = __builtin_amdgcn_readfirstlane(i / PU);
                    const int un = u0 + i - pp * PU; // the unit within its row; past the pitch: 0.0, as a row's own descriptor returns it
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
            if ((tid & 63) == 0) sh_slow[tid >> 6] = slow;
            if (tid == 0) sh_next = 0;
        }
        YSTAMP(11); // sweep: panel fill
        __syncthreads();
        unsigned fastmask = 0;
        {
            unsigned slow = 0;
#pragma unroll
            for (int w = 0; w < T / 64; w++) slow |= sh_slow[w];
            fastmask = ~slow;
        }
        // pending pivots whose pivot column lies in this panel (one panel in npanel per pending pivot): they patch an element (:25, :36)
        unsigned colmask = 0;
        for (int p = 0; p < npend; p++) {
            const int pcu = (__builtin_amdgcn_readfirstlane(pc[p]) >> 1) - u0;
            if ((unsigned)pcu < (unsigned)PU) colmask |= 1u << p;
        }
        // Two sets of D rows in flight per lane, A and B: the loads of both are issued before A is worked on, and by the time a
        // set's registers are loaded again its stores -- issued a whole set earlier -- have left.  (One set: the next batch's loads
        // reuse the registers the previous batch's stores read, hipcc waits for those stores to COMPLETE first, and a wave had
        // one batch of D x 16 bytes per lane per HBM write + read latency in flight: 2.7 TB/s at 16385^2.)
        const int r_any = tlist[0]; // (a valid slot for the unconditional reads of the rows that are not there)
        auto load_set = [&](int k0, double2 (&x)[D][U], int (&ri)[D], int (&rm)[D], int (&rp)[D]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int k = DYN ? k0 + d : k0 + d * RS + sub;
                ri[d] = k < nt ? tlist[k] : -1; // my row slot of each row in flight (-1: none)
                rm[d] = k < nt ? tmask[k] : -1;     // which pending pivots touch it (no row: whatever suits the others -- its registers are never stored) ...
                rp[d] = k < nt ? tpiv[k] : 0;       // ... and which have it as their pivot row
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * (ri[d] < 0 ? r_any : ri[d])) * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) x[d][u] = row_ld16<AUX>(rs, 16 * (u0 + lane + u * LU), 0);
            }
        };
        auto apply_set = [&](double2 (&x)[D][U], const int (&ri)[D], const int (&rm)[D], const int (&rp)[D]) __attribute__((always_inline)) {
            // The LDS reads of a pending pivot run ONE HALF AHEAD of the arithmetic: while the units [0, UH) of pending pivot p are
            // worked on, its units [UH, U) are on their way; while those are worked on, pivot p + 1's first half and its scalars
            // (my rows' coefficients, its column, its pivot-row slot).  All waves of a workgroup walk the pending pivots in step
            // behind the panel's barrier, so a read followed by the arithmetic that needs it was an LDS round trip nobody
            // covered -- twice per pending pivot: at 4097^2 a pending pivot cost a sweep 5.4 us where its arithmetic is 0.9.
            constexpr int UH = U > 4 ? U / 2 : U, NH = U / UH; // units of the pending row per half
            int rsl[D];
#pragma unroll
            for (int d = 0; d < D; d++) rsl[d] = ri[d] < 0 ? r_any : ri[d];
            const double *pan = panel + 2 * lane;
            auto rd_units = [&](int p, int ub, double2 (&pn)[UH]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < UH; u++) pn[u] = *reinterpret_cast<const double2 *>(pan + (size_t)p * 2 * PU + 2 * (ub + u) * LU);
            };
            auto rd_hdr = [&](int p, double (&cf)[D]) __attribute__((always_inline)) {
#pragma unroll
                for (int d = 0; d < D; d++) cf[d] = colv[p * rpw + rsl[d]];
            };
            auto work = [&](int p, const double (&cf_c)[D], int ub, const double2 (&pn_c)[UH]) __attribute__((always_inline)) {
                const int colxp = pc[p], lslotp = pl[p]; // (the rare path reads them where it needs them)
                const bool fastp = (fastmask >> p) & 1u;
                const int pcu = (colxp >> 1) - u0; // the pivot column's unit within this panel (uniform; in range or not)
                const bool col_here = (unsigned)pcu < (unsigned)PU;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const double coef = cf_c[d];
                    const bool piv = ri[d] == lslotp;
                    if (ri[d] < 0 || !(piv || fabs(coef) > 1e-16)) continue; // (uniform per wave) :31
                    if (fastp && !piv) { // (uniform) nothing of this panel's slice of the pivot row was flushed: two instructions per element
#pragma unroll
                        for (int u = 0; u < UH; u++) {
                            double2 &xv = x[d][ub + u];
                            const double px = coef * pn_c[u].x, py = coef * pn_c[u].y;
                            xv.x = xv.x - px;
                            xv.y = xv.y - py;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < UH; u++) {
                            double2 &xv = x[d][ub + u];
                            const double2 pn = pn_c[u];
                            const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
                            if (piv) {
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
                    if (col_here) { // (uniform; one panel in npanel) :25, :36 -- the one element of the row that the pivot column replaces
                        const double patch = nqv[p * rpw + ri[d]];
#pragma unroll
                        for (int u = 0; u < UH; u++)
                            if (pcu == lane + (ub + u) * LU) {
                                if (colxp & 1)
                                    x[d][ub + u].y = patch;
                                else
                                    x[d][ub + u].x = patch;
                            }
                    }
                }
            };
            // The usual pending pivot -- nothing of its row flushed in this panel, its column elsewhere, all my rows in flight touched
            // and none of them its pivot row -- is decided ONCE per trip from scalar bit masks and runs as straight-line code: two
            // fp64 instructions per element behind one scalar branch (the tests, row by row and half by half, were a third of a
            // pending pivot's instructions).  Everything else takes work() as before.
            unsigned plain = fastmask & ~colmask;
#pragma unroll
            for (int d = 0; d < D; d++) { // (the rows' masks were made with the list of touched rows, once per sweep)
                plain &= (unsigned)__builtin_amdgcn_readfirstlane(rm[d]) & ~(unsigned)__builtin_amdgcn_readfirstlane(rp[d]);
            }
            auto straight = [&](const double (&cf_c)[D], int ub, const double2 (&pn_c)[UH]) __attribute__((always_inline)) {
#pragma unroll
                for (int d = 0; d < D; d++)
#pragma unroll
                    for (int u = 0; u < UH; u++) {
                        double2 &xv = x[d][ub + u];
                        const double px = cf_c[d] * pn_c[u].x, py = cf_c[d] * pn_c[u].y;
                        xv.x = xv.x - px;
                        xv.y = xv.y - py;
                    }
            };
            double cfa[D], cfb[D];
            double2 pa[UH], pb[UH];
            rd_hdr(0, cfa);
            rd_units(0, 0, pa);
#pragma unroll 1
            for (int p = 0; p < npend; p++) {
                const int pnx = p + 1 < npend ? p + 1 : p; // (the last turn reads its own pivot again: nobody uses it)
                const bool is_plain = (plain >> p) & 1u;   // (scalar)
                if constexpr (NH == 2) {
                    rd_units(p, UH, pb);
                    __builtin_amdgcn_sched_barrier(0); // (the reads stay in front of the arithmetic they run ahead of)
                    if (is_plain)
                        straight(cfa, 0, pa);
                    else
                        work(p, cfa, 0, pa);
                    __builtin_amdgcn_sched_barrier(0);
                    rd_hdr(pnx, cfb);
                    rd_units(pnx, 0, pa);
                    __builtin_amdgcn_sched_barrier(0);
                    if (is_plain)
                        straight(cfa, UH, pb);
                    else
                        work(p, cfa, UH, pb);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    rd_hdr(pnx, cfb);
                    rd_units(pnx, 0, pb);
                    __builtin_amdgcn_sched_barrier(0);
                    if (is_plain)
                        straight(cfa, 0, pa);
                    else
                        work(p, cfa, 0, pa);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < UH; u++) pa[u] = pb[u];
                }
#pragma unroll
                for (int d = 0; d < D; d++) cfa[d] = cfb[d];
            }
        };
        auto store_set = [&](const double2 (&x)[D][U], const int (&ri)[D]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                if (ri[d] < 0) continue;
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * ri[d]) * pitch);
#pragma unroll
                for (int u = 0; u < U; u++) row_st16<AUX>(rs, 16 * (u0 + lane + u * LU), 0, x[d][u]);
            }
        };
        // The waves of a workgroup TAKE their rows, D at a time, from a counter in LDS as they get free: with a fixed share a wave whose
        // loads came back late kept everybody waiting at the next panel's barrier (stage stamps at 16385^2: 15 % of a sweep).
        // (ONE set of D rows in registers per wave: a second set -- the next trip's loads in flight under this trip's arithmetic -- was
        // built twice, as whole rows and as half-segments, and gained nothing: DESIGN.md 4.9c.  SETS stays in the signature for the record.)
        static_assert(SETS == 1 && LU == 64, "a wave takes D rows per trip");
        // (DYN = false: the fixed share -- same box, alternating runs, us per pivot taken / fixed: 6001^2 25.7 / 26.5, 8193^2 32.8 / 32.95,
        // 4097^2 18.7 / 18.6, but 16385^2 79.7 / 76.8: the callers keep the fixed share for rows of 16 units per lane)
#pragma unroll 1
        for (int kf = 0;; kf += RS * D) {
            int k0 = kf;
            if constexpr (DYN) {
                if ((tid & 63) == 0) k0 = __hip_atomic_fetch_add(&sh_next, D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                k0 = __builtin_amdgcn_readlane(k0, 0);
            }
            if (k0 >= nt) break; // (uniform)
            double2 xa[D][U];
            int ria[D], rma[D], rpa[D];
            YSTAMP(12); // sweep: barrier behind the fill, flags / between trips
            load_set(k0, xa, ria, rma, rpa);
#ifdef YALPS_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            YSTAMP(13); // sweep: my rows' loads (diagnostic build waits for them here)
            apply_set(xa, ria, rma, rpa);
            YSTAMP(14); // sweep: the pending pivots applied in registers
            store_set(xa, ria);
            YSTAMP(15); // sweep: stores issued
        }
    }
    if (ntail > 0) { // (uniform)
        const int ut0 = npanel * PU; // first unit of the tail
        constexpr int PC = 8;        // pending rows' units of a lane in flight at once
#pragma unroll 1
        for (int it = tid; it < nt * ntail; it += T) {
            const int k = it / ntail, ut = ut0 + (it - k * ntail), ri = tlist[k];
            double2 *px = reinterpret_cast<double2 *>(mat + (size_t)(b + NB * ri) * pitch) + ut;
            double2 xv = *px;
#pragma unroll 1
            for (int p0 = 0; p0 < npend; p0 += PC) {
                double2 pn_c[PC];
#pragma unroll
                for (int q = 0; q < PC; q++)
                    if (p0 + q < npend) pn_c[q] = *(reinterpret_cast<const double2 *>(pend0 + (size_t)(p0 + q) * pitch) + ut);
#pragma unroll
                for (int q = 0; q < PC; q++) {
                    const int p = p0 + q;
                    if (p >= npend) break; // (uniform)
                    const double coef = colv[p * rpw + ri];
                    const bool piv = ri == pl[p];
                    if (!(piv || fabs(coef) > 1e-16)) continue; // :31
                    const double2 pn = pn_c[q];
                    const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
                    const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
                    if (piv) {
                        xv.x = f0 ? pn.x : 0.0;
                        xv.y = f1 ? pn.y : 0.0;
                    } else {
                        const double px_ = coef * pn.x, py_ = coef * pn.y;
                        const double nx = xv.x - px_, ny = xv.y - py_;
                        xv.x = f0 ? nx : xv.x;
                        xv.y = f1 ? ny : xv.y;
                    }
                    const int colxp = pc[p];
                    if ((colxp >> 1) == ut) { // :25, :36 -- the one element of the row that the pivot column replaces
                        const double patch = nqv[p * rpw + ri];
                        if (colxp & 1)
                            xv.y = patch;
                        else
                            xv.x = patch;
                    }
                }
            }
            *px = xv;
        }
    }
    YSTAMP(16); // sweep: tail units
    __syncthreads(); // (the panel LDS may be reused by the caller)
    YSTAMP(17); // sweep: last barrier
}

// ------------------------------------------------------------------------------------------
// direct_flush: the same sweep WITHOUT the LDS panels -- round 2's form: RB (half-)rows of JH units per lane in registers, the
// pending rows' units read from L2 once per RB rows.  For workgroups with few rows (a rank's share of a row-sharded
// tableau: 8 rows per workgroup at 2049 x 16385) the panels cost more than they save: per panel two barriers and an LDS fill
// of npend x 8 KB against 8 rows x 8 KB of row data (measured: 46 -> 53 us per pivot with panels there, 144 -> 107 at
// 64 rows per workgroup).  The host picks by rows per workgroup (yalps_hip.hip, DSHARD_PANEL_MIN_ROWS).
// Lane `tid` holds units tid, tid + T, ... of a row; lane_off = 16 * tid.
// ------------------------------------------------------------------------------------------
template <int T, int J, int RB, bool NT, bool PF = true, typename RsrcOf>
__device__ __forceinline__ void direct_flush(double *mat, int pitch, int b, int NB, const double *pend0, int npend, const double *colv,
                                             const double *nqv, int rpw, const int *pl, const int *pc, const int *tlist, int nt, RsrcOf rsrc_of) {
    constexpr int JH = J > 8 ? 8 : J, JA = JH; // (RB (half-)rows of JH units per lane in registers)
    constexpr int AUX = NT ? AUX_NT : AUX_PLAIN;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane_off = 16 * tid;
#pragma unroll 1
    for (int u0 = 0; u0 < J; u0 += JH) {
#pragma unroll 1
        for (int k = 0; k < nt; k += RB) {
            double2 xb[RB][JH];
            int ri[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                ri[u] = tlist[k + u < nt ? k + u : k];
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * ri[u]) * pitch);
                if (k + u < nt) {
#pragma unroll
                    for (int j = 0; j < JH; j++) xb[u][j] = row_ld16<AUX>(rs, lane_off + 16 * T * (u0 + j), 0);
                }
            }
            const int cnt = nt - k;
            // (the L2 reads of pending row p + 1 are in flight while row p is applied: a read followed by the arithmetic on it was one
            // L2 round trip per pending pivot and batch that nothing covered -- the waves of a workgroup walk the pending pivots in step)
            auto rd_pend = [&](int p, double2 (&pn)[JA]) __attribute__((always_inline)) {
                const __amdgpu_buffer_rsrc_t rsp = rsrc_of(pend0 + (size_t)p * pitch);
#pragma unroll
                for (int j = 0; j < JA; j++) pn[j] = row_ld16<AUX_PLAIN>(rsp, lane_off + 16 * T * (u0 + j), 0);
            };
            auto apply_pend = [&](int p, const double2 (&pn)[JA]) __attribute__((always_inline)) {
                const int colxp = pc[p], lslotp = pl[p];
                // the one element of a row that the pivot column replaces (:25, :36): unit `up` of lane `lp`
                const int up = (colxp >> 1) / T;
                const bool lane_p = ((colxp >> 1) % T) == tid;
                double coefu[RB], patchu[RB];
                bool pivu[RB], actu[RB];
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    coefu[u] = colv[p * rpw + ri[u]];
                    patchu[u] = nqv[p * rpw + ri[u]];
                    pivu[u] = ri[u] == lslotp;
                    actu[u] = u < cnt && (pivu[u] || fabs(coefu[u]) > 1e-16); // :31
                }
                bool fl = false; // (per wave: nothing of these units was flushed -> the select-free path)
#pragma unroll
                for (int j = 0; j < JA; j++)
                    fl = fl || (unsigned long long)__double_as_longlong(pn[j].x) == FLUSHED || (unsigned long long)__double_as_longlong(pn[j].y) == FLUSHED;
                const bool fastp = __builtin_amdgcn_ballot_w64(fl) == 0;
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    if (!actu[u]) continue; // (uniform)
#pragma unroll
                    for (int j = 0; j < JA; j++) {
                        double2 &xv = xb[u][j];
                        if (fastp && !pivu[u]) {
                            const double px = coefu[u] * pn[j].x, py = coefu[u] * pn[j].y;
                            xv.x = xv.x - px;
                            xv.y = xv.y - py;
                        } else {
                            const bool f0 = (unsigned long long)__double_as_longlong(pn[j].x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pn[j].y) != FLUSHED;
                            if (pivu[u]) {
                                xv.x = f0 ? pn[j].x : 0.0;
                                xv.y = f1 ? pn[j].y : 0.0;
                            } else {
                                const double px = coefu[u] * pn[j].x, py = coefu[u] * pn[j].y;
                                const double nx = xv.x - px, ny = xv.y - py;
                                xv.x = f0 ? nx : xv.x;
                                xv.y = f1 ? ny : xv.y;
                            }
                        }
                        if (up == u0 + j) { // (uniform)
                            if (lane_p) {
                                if (colxp & 1)
                                    xv.y = patchu[u];
                                else
                                    xv.x = patchu[u];
                            }
                        }
                    }
                }
            };
            if constexpr (PF) {
                double2 pna[JA], pnb[JA];
                rd_pend(0, pna);
#pragma unroll 1
                for (int p = 0; p < npend; p += 2) {
                    const int p1 = p + 1 < npend ? p + 1 : p, p2 = p + 2 < npend ? p + 2 : p; // (past the end: a row read again, nobody uses it)
                    rd_pend(p1, pnb);
                    __builtin_amdgcn_sched_barrier(0);
                    apply_pend(p, pna);
                    __builtin_amdgcn_sched_barrier(0);
                    rd_pend(p2, pna);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < npend) apply_pend(p + 1, pnb); // (uniform)
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 1
                for (int p = 0; p < npend; p++) {
                    double2 pn[JA];
                    rd_pend(p, pn);
                    apply_pend(p, pn);
                }
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                if (k + u < nt) {
                    const __amdgpu_buffer_rsrc_t rs = rsrc_of(mat + (size_t)(b + NB * ri[u]) * pitch);
#pragma unroll
                    for (int j = 0; j < JH; j++) row_st16<AUX>(rs, lane_off + 16 * T * (u0 + j), 0, xb[u][j]);
                }
            }
        }
    }
}
