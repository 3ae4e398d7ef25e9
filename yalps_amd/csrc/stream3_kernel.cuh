// stream3_kernel.cuh -- stream2_kernel (delayed row updates, several pivots per sweep) for rows too wide for it
// Part of libyalps_hip.so; included by persistent_stream3.hip inside its unnamed namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// stream3_kernel<T lanes, J units per lane per row, NT>: the same algorithm as stream2_kernel -- a pivot's elimination
// stays pending, the next decisions are made from scalars, every touched row is streamed once per d.delay_depth pivots and
// gets all pending eliminations in registers (bit for bit what that many sweeps leave) -- for rows of 8194 .. 16385
// columns (BASELINE config 5), where neither a lane's 16 units of the objective replica nor two normalised pivot rows of
// 131 KB fit where stream2_kernel keeps them.  Placement:
//   * the objective replica lives in registers (2 J doubles per lane: the columns the lane also holds of a row; round 2 kept
//     it in LDS, which the sweep's panels of pending rows need now -- panel_flush.cuh -- and the sweep no longer needs the registers);
//   * the pending normalised pivot rows live in a scratch in global memory shared by the workgroups of an XCD (d.pend: [8 XCDs]
//     [2 sets][depth][pitch], <= 2 MB per XCD: resident in its L2): written when the pivot is decided (plain stores, by the lane that will
//     read them during the sweep; every workgroup stores the same bytes), read 16 bytes per lane, unit and pending pivot per
//     two rows during the sweep, single entries for the scalar chains with agent-scope loads;
//   * a pivot row passes through registers 8 units per lane at a time (normalise, objective replica, pricing in one pass).
// ------------------------------------------------------------------------------------------
template <int T, int J, bool NT, bool CHECK = false, bool PANEL = true>
__global__ __launch_bounds__(T) void stream3_kernel(Desc d, int parity, int chunk) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_q, sh_c0; // quotient; objective-row entry of the pivot column
    __shared__ int sh_fail, sh_nt, sh_flag, sh_verdict;
    __shared__ double sh_rowrhs;
    // two-step exchange, round 3: the winner's row is materialised by ALL workgroups of its owner's XCD in column slices
    __shared__ unsigned char sh_xcc[256]; // everybody's XCD (1 + HW_REG_XCC_ID), read once per launch from d.hp_xcc
    __shared__ double sh_hs[HP_SCAL];     // the winner's row's scalars as its owner published them (layout: publish())
    __shared__ int sh_hk[2];              // my index among the workgroups of my XCD, and their number
    constexpr int MAXD = STREAM3_MAXD;
    constexpr int PU = stream3_panel_units(J); // 16-byte units of a row per LDS panel of the sweep (panel_flush.cuh)
    // TWO: the exchange in two steps -- every workgroup publishes its candidate's KEY only; the workgroup that owns the winner
    // then publishes that one row (pending pivots applied) behind a second record.  One more hand-off on the pivot's chain,
    // but 1 row instead of 256 goes through the pending pivots and the L2s per pivot (at 16 units per lane that was 335 MB of
    // L2 traffic per pivot).  Rows of 6 units per lane and more (4098+ columns; 5001^2 27.0 -> 25.7 us per pivot, 16385^2 167 -> 160);
    // at 4 units (4097^2) it changes nothing measurable: the single hand-off stays there.
#ifndef YALPS_S3_TWO_MIN_J
#define YALPS_S3_TWO_MIN_J 6
#endif
    constexpr bool TWO = J >= YALPS_S3_TWO_MIN_J && !CHECK; // (with hasCycle the 16-unit form does not fit the registers)
    __shared__ int sh_pl[MAXD], sh_pc[MAXD]; // the pending pivots, oldest first: my slot of the pivot row (-1: not mine), pivot column (mat index)
    __shared__ int sh_fast[MAXD][T / 64];       // per wave: nothing of its slice of that pivot row was flushed (:31 select-free path)
    constexpr int JC = J > 8 ? 8 : J; // units per lane that pass through registers at a time (a pivot row being decided)
    extern __shared__ __attribute__((aligned(16))) double sm_dyn[]; // colv[depth][rpw], nqv[depth][rpw], lav[rpw], rhsv[rpw], tlist[rpw] (int), panel[depth][2 PU]

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
    int64_t hist_len = Sin->hist_len; // checkCycles: pivots recorded in the current phase (src/simplex.ts:67,107)
    int slot = 0;
    const int rpw = (d.hcap + NB - 1) / NB;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    const int depth = d.delay_depth < 1 ? 1 : d.delay_depth > MAXD ? MAXD : d.delay_depth;
    double *colv0 = sm_dyn, *nqv0 = colv0 + (size_t)depth * rpw, *lav = nqv0 + (size_t)depth * rpw,
           *rhsv = lav + rpw; // (nqv: what replaces a row's pivot-column entry, :25 / :36 -- one division per row and pivot, by one lane)
    int *tlist = reinterpret_cast<int *>(rhsv + rpw), *tmask = tlist + (rpw + 3) / 4 * 4, *tpiv = tmask + (rpw + 3) / 4 * 4;
    double *panel = rhsv + rpw + (rpw + 3) / 4 * 6; // (behind tlist, tmask and tpiv, 16-byte aligned)
    const double flushed = __longlong_as_double((long long)FLUSHED);
    // The pending normalised pivot rows: one scratch PER XCD (d.pend: [8 XCDs][2 sets][depth][pitch]).  Every workgroup computes the
    // same rows from the same published bytes; the workgroups of one XCD store them to the same place (identical values;
    // a lane only ever reads back columns it stores itself, or, for the scalar chains, columns stored before its workgroup's
    // barrier) and read them out of the L2 they share.  Private copies per workgroup (256 x depth x 131 KB) did not stay in
    // the L2s; ONE copy for all XCDs is wrong with write-back stores -- the L2s are not coherent with each other, a dirty
    // copy of an address from an earlier sweep generation lingering in another XCD's L2 was written back OVER the fresh row
    // after this XCD had evicted it (tools/soak_delay.py, a 2 % dense tableau, case 1147: one workgroup's three touched rows
    // wrong in the pivot row's 54 non-zero columns, not reproducible run to run) -- and costs 3-18 % with write-through
    // stores.  The XCD is read from the hardware register: what matters is that workgroups with the same id share an L2,
    // not which id a workgroup gets.  Two sets, taken in turns from sweep to sweep: a workgroup that is through with its
    // sweep may store the next pivot's row while another still sweeps (it cannot get further: deciding the pivot after that
    // needs every workgroup's next candidate, published after its sweep).
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    double *const pend_xcd = d.pend + (size_t)(xcc & 7) * 2 * depth * pitch;
    double *prow0 = pend_xcd;

    // (rows and scratch rows are addressed through a buffer descriptor of ONE row + a 32-bit lane offset, like sweep_kernel:
    // flat addressing held a 64-bit pair per unit, pivot and row in flight -- 192 spilled registers)
    const int lane_off = 16 * tid, row_bytes = pitch * 8;
    auto rsrc_of = [&](const double *row_ptr) __attribute__((always_inline)) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(row_ptr);
        const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(u), 0, row_bytes, 0x00020000);
    };
    unsigned padmask = 0; // columns of mine that do not exist (c0 + k >= n): 0.0 in a pivot row, must not count as "flushed"
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (2 * (tid + j * T) + k >= (CHECK ? pitch : n)) padmask |= 1u << (2 * j + k); // (CHECK: see the padding note at the end of the pivot loop)
    constexpr unsigned FULL = J == 16 ? 0xFFFFFFFFu : (1u << (2 * (J & 15))) - 1u;
    // ---- my replica of the objective row (registers), my rows' RHS (LDS) ----
    double2 ob[J]; // my replica of the objective row: units tid, tid + T, ...
#pragma unroll
    for (int j = 0; j < J; j++) ob[j] = row_ld16<AUX_PLAIN>(rsrc_of(mat), lane_off + 16 * T * j, 0);
    for (int i = tid; i < my_rows; i += T) rhsv[i] = rhs[b + NB * i];
    if (tid == 0) sh_fail = 0;
    if constexpr (TWO) { // my XCD, for everybody: out before my first key record
        if (tid == 0) {
            __hip_atomic_store(d.hp_xcc + b, (xcc & 7) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();

    // ---- the pending pivots (npend of them, [0] the oldest): scalars, pivot rows and my rows' pivot-column entries in LDS ----
    int npend = 0, pset = 0; // (pset: which of the two scratch sets holds the rows pending now)
    // entry (my row slot i, mat column c) after ONE pending pivot, given the entry before it (:14-25, :31-36 for one element)
    auto after1 = [&](double p, const double *colvp, const double *nqvp, int lslotp, int colxp, int i, double v, int c)
                      __attribute__((always_inline)) {
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
        int t0 = tid;
        asm volatile("" : "+v"(t0)); // (opaque: my rows' base addresses are recomputed here, not hoisted out of the pivot loop and spilled)
        for (int i = t0; i < my_rows; i += T) {
            // (the entry and the pending pivot rows' entries of that column: all loads in flight at once -- one after the
            // other they were a dependent trip through L2 per pending pivot on every pivot's chain)
            double v = __hip_atomic_load(mat + (size_t)(b + NB * i) * pitch + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
            for (int p0 = 0; p0 < npend; p0 += 8) { // (eight pending pivots' entries in flight at a time: registers)
                double pe[8];
#pragma unroll
                for (int p = 0; p < 8; p++)
                    pe[p] = p0 + p < npend ? __hip_atomic_load(prow0 + (size_t)(p0 + p) * pitch + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                for (int p = 0; p < 8; p++)
                    if (p0 + p < npend) v = after1(pe[p], colv0 + (p0 + p) * rpw, nqv0 + (p0 + p) * rpw, sh_pl[p0 + p], sh_pc[p0 + p], i, v, c);
            }
            out[i] = v;
        }
        __syncthreads();
    };
    // the pending pivots applied to (up to) RB half-rows held in registers: units [u0, u0 + JH) of my row slots ri[0 .. cnt)
    // (round 3: only the candidate / winner's row that is about to be published goes this way -- one row, JH units of it per lane at
    // a time, the pending rows read from my XCD's scratch; the sweep of all rows is panel_flush.cuh)
    constexpr int JH = J > 8 ? 4 : J, RB = 1; // (16-unit rows: four units at a time, the units of four pending pivots in flight beside them)
    auto apply_batch = [&](int u0, double2 (&xb)[RB][JH], const int (&ri)[RB], int cnt) __attribute__((always_inline)) {
        // The pending rows' units come from my XCD's scratch (L2).  The row about to be published is alone on the pivot's chain
        // here, so what counts is the number of dependent L2 round trips: the units of G pending pivots are loaded at once
        // (G x JH 16-byte loads per lane in flight) and applied one pivot after the other (round 2: one pivot per trip, JA units:
        // 2 x npend trips at 16 units per lane -- 11.7 us with 8 pending, twice that with 16).
        constexpr int G = JH >= 8 ? 2 : JH >= 4 ? 4 : 8; // pending pivots per trip: 16 loads per lane in flight
#pragma unroll 1
        for (int p0 = 0; p0 < npend; p0 += G) {
            double2 pn[G][JH];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int p = p0 + g < npend ? p0 + g : p0; // (uniform)
                const __amdgpu_buffer_rsrc_t rsp = rsrc_of(prow0 + (size_t)p * pitch);
#pragma unroll
                for (int j = 0; j < JH; j++) pn[g][j] = row_ld16<AUX_PLAIN>(rsp, lane_off + 16 * T * (u0 + j), 0);
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int p = p0 + g;
                if (p >= npend) break; // (uniform)
                const int colxp = sh_pc[p], lslotp = sh_pl[p];
                const bool fastp = sh_fast[p][tid >> 6] != 0;
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    const double coefu = colv0[p * rpw + ri[u]], patchu = nqv0[p * rpw + ri[u]];
                    const bool pivu = ri[u] == lslotp;
                    if (!(u < cnt && (pivu || fabs(coefu) > 1e-16))) continue; // (uniform) :31
#pragma unroll
                    for (int j = 0; j < JH; j++) {
                        const int c0 = 2 * (tid + (u0 + j) * T);
                        double2 &xv = xb[u][j];
                        const double2 pv_ = pn[g][j];
                        if (fastp && !pivu) {
                            const double px = coefu * pv_.x, py = coefu * pv_.y;
                            xv.x = xv.x - px;
                            xv.y = xv.y - py;
                        } else {
                            const bool f0 = (unsigned long long)__double_as_longlong(pv_.x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pv_.y) != FLUSHED;
                            if (pivu) {
                                xv.x = f0 ? pv_.x : 0.0;
                                xv.y = f1 ? pv_.y : 0.0;
                            } else {
                                const double px = coefu * pv_.x, py = coefu * pv_.y;
                                const double nx = xv.x - px, ny = xv.y - py;
                                xv.x = f0 ? nx : xv.x;
                                xv.y = f1 ? ny : xv.y;
                            }
                        }
                        if (c0 == (colxp & ~1)) {
                            if (colxp & 1)
                                xv.y = patchu;
                            else
                                xv.x = patchu;
                        }
                    }
                }
            }
        }
    };
    // every touched row streamed once, all pending eliminations in registers; afterwards nothing is pending
#ifdef YALPS_STAMPS
    unsigned long long st_acc[20] = {}, st_last = 0, st_t0 = 0, st_r0 = 0; // (diagnostic build; the sweep's own stages are summed inside panel_flush)
#endif
    auto flush_pending = [&]() __attribute__((always_inline)) {
        if (npend == 0) return; // (uniform)
        int tl = tid;
        asm volatile("" : "+v"(tl)); // (opaque: the lane masks below are not kept across the pivot loop)
        if (tl < 64) { // compact list of my touched rows (wave 0)
            int cnt = 0;
            for (int base = 0; base < my_rows; base += 64) {
                const int i = base + tl;
                bool t = false;
                int msk = 0, pvm = 0; // bit p of msk: pending pivot p touches the row (:31, or its pivot row); of pvm: the row is p's pivot row
                if (i < my_rows && b + i > 0) { // (not the objective row: the replica of it IS that row with every pending pivot applied)
#pragma unroll 4
                    for (int p = 0; p < npend; p++) { // (no short circuit: the LDS reads of four pending pivots in flight, not a chain of round trips)
                        const int pv = i == sh_pl[p] ? 1 : 0, tc = (pv | (fabs(colv0[p * rpw + i]) > 1e-16 ? 1 : 0));
                        msk |= tc << p;
                        pvm |= pv << p;
                    }
                    t = msk != 0;
                }
                const unsigned long long m = __ballot(t);
                if (t) {
                    const int k = cnt + __popcll(m & ((1ull << tl) - 1ull));
                    tlist[k] = i;
                    tmask[k] = msk;
                    tpiv[k] = pvm;
                }
                cnt += __popcll(m);
            }
            if (tl == 0) sh_nt = cnt;
        }
        __syncthreads();
        // (panel_flush.cuh: the pending rows staged in LDS one 1024-column panel at a time, a wave per row, eight units per lane, two rows in flight per wave)
        // Wide rows: the objective replica's 4 J registers per lane are parked in global memory for the duration of the sweep (2 x
        // 8 w bytes per workgroup and sweep against 16 w bytes per ROW): with them live, the 16-unit forms spill in the sweep.
        constexpr bool PARK = J >= 6;
        // The objective row is not swept: workgroup 0 stores its replica -- the same arithmetic, pivot by pivot (:27-38 for row 0) -- over it.
        // (With it the first workgroup had one row more than the others wherever the row count is 2^k + 1 -- every BASELINE
        // configuration: a fifth trip of 8 waves x 2 rows per panel for ONE row at 16385^2, a second one at 4097^2, everybody waiting.)
        if (b == 0) {
            const __amdgpu_buffer_rsrc_t r0 = rsrc_of(mat);
#pragma unroll
            for (int j = 0; j < J; j++) row_st16<NT ? AUX_NT : AUX_PLAIN>(r0, lane_off + 16 * T * j, 0, ob[j]);
        }
        if constexpr (PARK) {
            const __amdgpu_buffer_rsrc_t rpk = rsrc_of(d.ob_park + (size_t)b * pitch);
#pragma unroll
            for (int j = 0; j < J; j++) row_st16<AUX_PLAIN>(rpk, lane_off + 16 * T * j, 0, ob[j]);
        }
        if constexpr (PANEL)
            panel_flush<T, PU, 64, YALPS_PANEL_D, YALPS_PANEL_SETS, NT, 4, (J < 16)>(mat, pitch, b, NB, prow0, npend, colv0, nqv0, rpw, sh_pl, sh_pc, tlist, tmask, tpiv, sh_nt, panel, rsrc_of YSTAMP_ARGS); // (4 loads of the fill in flight: with 8 hipcc spills loop invariants of the pivot loop in some instantiations; J < 16: the waves take their rows from a counter)
        else // (few rows per workgroup: the pending rows straight from my XCD's scratch, round 2's form -- the panels' barriers and LDS
             // fills cost more than they save there: 1025 x 16385, 4 rows per workgroup, 32 -> 38 us per pivot with panels)
            direct_flush<T, J, (J >= 8 ? 2 : 3), NT, !(J == 16 && CHECK)>(mat, pitch, b, NB, prow0, npend, colv0, nqv0, rpw, sh_pl, sh_pc, tlist, sh_nt, rsrc_of);
        if constexpr (PARK) { // (same lane, same addresses: the stores above are ordered in front of these loads)
            const __amdgpu_buffer_rsrc_t rpk = rsrc_of(d.ob_park + (size_t)b * pitch);
#pragma unroll
            for (int j = 0; j < J; j++) ob[j] = row_ld16<AUX_PLAIN>(rpk, lane_off + 16 * T * j, 0);
        }
        npend = 0;
        pset ^= 1;
        prow0 = pend_xcd + (size_t)pset * depth * pitch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // my rows are complete in memory (and the LDS arrays free) before anything reads them again
    };

    int la = 0; // entering column of the NEXT pivot (phase 2), priced on my objective replica
    auto price = [&]() __attribute__((always_inline)) { // src/simplex.ts:71-79
        KI best = {INFINITY, INT_MAX};
        int tp = tid;
        asm volatile("" : "+v"(tp)); // (opaque: the 2 J column numbers of a lane are recomputed here, not kept across the pivot loop)
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tp + j * T);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double ov = elem(ob[j], k);
                if (c0 + k < n && ov > precision && ki_better(-ov, c0 + k + 1, best.k, best.i)) {
                    best.k = -ov;
                    best.i = c0 + k + 1;
                }
            }
        }
        best = block_argmin<T>(best, sk, si, slot);
        slot ^= 1;
        la = __builtin_amdgcn_readfirstlane(best.i == INT_MAX ? 0 : best.i); // (uniform: kept in a scalar register)
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
    [[maybe_unused]] bool xcc_ready = false;
    // a row of mine AS IT IS NOW (memory + pending pivots, in registers; not stored in place) -> my slot of the hand-off rows: write-through, drained
    auto publish_row = [&](int par, int cg) __attribute__((always_inline)) {
        if (my_rows > 0) {
            const __amdgpu_buffer_rsrc_t rsm = rsrc_of(mat + (size_t)(b + NB * cg) * pitch), rsd = rsrc_of(d.rc_rows[par] + (size_t)b * pitch);
    #pragma unroll 1
            for (int u0 = 0; u0 < J; u0 += JH) {
                double2 xb[RB][JH];
                int ri[RB];
#pragma unroll
                for (int u = 0; u < RB; u++) ri[u] = cg;
#pragma unroll
                for (int j = 0; j < JH; j++) xb[0][j] = row_ld16<AUX_PLAIN>(rsm, lane_off + 16 * T * (u0 + j), 0);
                apply_batch(u0, xb, ri, 1);
#pragma unroll
                for (int j = 0; j < JH; j++) row_st16<AUX_SC1>(rsd, lane_off + 16 * T * (u0 + j), 0, xb[0][j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
        __syncthreads();                                  // ... before ONE lane raises the flag
    };
    // my candidate: (single hand-off) with its row; (TWO) its key only
    auto publish = [&](KI cand) __attribute__((always_inline)) {
        epoch++;
        const int par = epoch & 1, cg = cand.i == INT_MAX ? 0 : cand.i / NB;
        if constexpr (!TWO) {
            if (tid == 0) st_sc1(d.rc_key[par] + b, rhsv[cg]); // the candidate row's RHS entry
            publish_row(par, cg);
        } else {
            // TWO: with the key travel the candidate row's SCALARS -- its RHS entry, and per pending pivot its pivot-column entry as it
            // was, what replaces it, whether it was that pivot's pivot row -- so that, should this row win, the workgroups of my XCD
            // can run it through the pending pivots in column slices (HP_SCAL doubles, write-through, drained before the record)
            if (tid < HP_SCAL) {
                double v = 0.0;
                if (tid == 0) {
                    int off = (2 * depth + 1) * rpw; // (= rhsv - sm_dyn, worked out here in a scalar register: kept as a pointer across the pivot loop it was the one value hipcc spilled)
                    asm volatile("" : "+s"(off));
                    v = sm_dyn[off + cg];
                } else if (tid == 1) {
                    int m = 0;
                    for (int p = 0; p < npend; p++) m |= (sh_pl[p] == cg ? 1 : 0) << p;
                    v = (double)m;
                } else if (tid >= 8 && tid < 8 + 2 * MAXD) { // (colv0 / nqv0 = sm_dyn + 0 / + depth * rpw: the offset worked out here in a scalar register, as above)
                    const int q = tid - 8, pq = q < MAXD ? q : q - MAXD;
                    int off = q < MAXD ? 0 : depth * rpw;
                    int base0 = 0;
                    asm volatile("" : "+s"(base0));
                    if (pq < npend) v = sm_dyn[base0 + off + pq * rpw + cg];
                }
                st_sc1(d.hp_scal + ((size_t)par * NB + b) * HP_SCAL + tid, v);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
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
#ifdef YALPS_STAMPS
    // diagnostic build: stage sums over the launch's pivots (stages: tools/stream3_stages.py)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0)::"memory");
    st_last = st_t0;
#endif

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
        YSTAMP(0); // everybody's key record polled, arg-min (barrier)
        if (c.i == INT_MAX) {
            if (phase == 1) { // :120 phase 1 is over: same tableau, now the min-ratio exchange
                phase = 2;
                iter = 0.0;
                hist_len = 0;
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
        const int row_in = __builtin_amdgcn_readfirstlane(c.i), row = (unsigned)row_in < (unsigned)h ? row_in : 0, owner = row % NB; // (uniform: scalar registers)
        if (row != row_in && tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (never expected)
        const int lslot = owner == b ? row / NB : -1; // my slot of the pivot row, if I own it
        // ---------------- the winner's row as published: the tableau's row after every earlier pivot ----------------
        const double *src = d.rc_rows[par] + (size_t)owner * pitch;
        double rhs_row;
        if constexpr (TWO) {
            // The winner's row, as it is now, into rc_rows[par][owner]: every workgroup of its OWNER'S XCD takes a slice of its
            // columns -- they share the L2 that holds the owner's rows and that XCD's copy of the pending rows --, loads the row's
            // units and the pending rows' units (all loads in flight at once), applies the pending pivots with the scalars the owner
            // published with its key, stores the slice write-through and raises its own flag; everybody waits for those flags.
            // (Round 2: the owner alone, (1 + npend) x 131 KB through one CU at ~50 GB/s: 11.7 us at 16385 columns with 8
            // pending, 23 us with 16 -- the largest item of a pivot's head.)
            if (!xcc_ready) { // once per launch: everybody's XCD (published before the first key record, which I have seen by now)
                if (tid < NB) sh_xcc[tid] = (unsigned char)__hip_atomic_load(d.hp_xcc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (tid == 0) {
                    int k = 0, K = 0;
                    for (int w_ = 0; w_ < NB; w_++) {
                        if (sh_xcc[w_] == sh_xcc[b]) {
                            if (w_ < b) k++;
                            K++;
                        }
                    }
                    sh_hk[0] = k;
                    sh_hk[1] = K;
                }
                __syncthreads();
                xcc_ready = true;
            }
            const int oxcc = sh_xcc[owner];
            const double *sc = d.hp_scal + ((size_t)par * NB + owner) * HP_SCAL;
            if (oxcc == sh_xcc[b]) { // (uniform) I am one of the helpers
                if (tid < HP_SCAL) sh_hs[tid] = ld_sc1(sc + tid);
                __syncthreads();
                const int hk = sh_hk[0], hK = sh_hk[1], units = pitch >> 1;
                const int per = (units + hK - 1) / hK, u_lo = hk * per, u_hi = u_lo + per < units ? u_lo + per : units;
                const int pivmask = (int)sh_hs[1];
                const double *rowp = mat + (size_t)row * pitch;
                double *dstp = d.rc_rows[par] + (size_t)owner * pitch;
                for (int u = u_lo + tid; u < u_hi; u += T) {
                    double2 xv = ld16_sc1_one(rowp + 2 * u); // (sc1: past my L1 -- the row is another workgroup's, written in its sweep)
#pragma unroll 1
                    for (int p0 = 0; p0 < npend; p0 += 8) {
                        double2 pn[8];
#pragma unroll
                        for (int g = 0; g < 8; g++)
                            pn[g] = p0 + g < npend ? ld16_sc1_one(prow0 + (size_t)(p0 + g) * pitch + 2 * u) : make_double2(0.0, 0.0);
#pragma unroll
                        for (int g = 0; g < 8; g++) {
                            const int p = p0 + g;
                            if (p >= npend) break; // (uniform)
                            const double coef = sh_hs[8 + p];
                            const bool piv = (pivmask >> p) & 1;
                            if (!(piv || fabs(coef) > 1e-16)) continue; // :31
                            const bool f0 = (unsigned long long)__double_as_longlong(pn[g].x) != FLUSHED;
                            const bool f1 = (unsigned long long)__double_as_longlong(pn[g].y) != FLUSHED;
                            if (piv) {
                                xv.x = f0 ? pn[g].x : 0.0;
                                xv.y = f1 ? pn[g].y : 0.0;
                            } else {
                                const double px = coef * pn[g].x, py = coef * pn[g].y;
                                const double nx = xv.x - px, ny = xv.y - py;
                                xv.x = f0 ? nx : xv.x;
                                xv.y = f1 ? ny : xv.y;
                            }
                            const int colxp = sh_pc[p];
                            if (2 * u == (colxp & ~1)) { // :25, :36
                                if (colxp & 1)
                                    xv.y = sh_hs[8 + MAXD + p];
                                else
                                    xv.x = sh_hs[8 + MAXD + p];
                            }
                        }
                    }
                    st16_sc1(dstp + 2 * u, xv);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
                __syncthreads();                                  // ... before ONE lane raises my flag
                if (tid == 0)
                    __hip_atomic_store(d.hp_flag + (size_t)par * NB + b, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < NB && sh_xcc[tid] == oxcc) { // everybody: the flags of the owner's XCD
                unsigned spins = 0;
                unsigned long long spin_t0 = 0;
                for (;;) {
                    if (__hip_atomic_load(d.hp_flag + (size_t)par * NB + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)epoch) break;
                    if (spin_expired(spins, spin_t0, d.rc_err)) {
                        sh_fail = 1;
                        __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (tid == 0) sh_rowrhs = ld_sc1(sc); // the winner's row's RHS entry
            __syncthreads();
            if (sh_fail) return;
            rhs_row = sh_rowrhs;
        } else {
            rhs_row = ld_sc1(d.rc_key[par] + owner);
        }
        YSTAMP(1); // two-step exchange: the winner's owner runs the row through the pending pivots and publishes it / the others wait for its record
        const __amdgpu_buffer_rsrc_t rsrc_src = rsrc_of(src);
        int col = la;
        if (phase == 1) { // :123-134, JC units of the raw row per lane at a time
            KI e = {INFINITY, INT_MAX};
            int tp = tid;
            asm volatile("" : "+v"(tp)); // (opaque: see price())
#pragma unroll
            for (int jb = 0; jb < J; jb += JC) {
                double2 pv[JC];
#pragma unroll
                for (int j = 0; j < JC; j++) pv[j] = row_ld16<AUX_SC1>(rsrc_src, lane_off + 16 * T * (jb + j), 0);
#pragma unroll
                for (int j = 0; j < JC; j++) {
                    const int c0 = 2 * (tp + (jb + j) * T);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const double coefficient = elem(pv[j], k);
                        if (c0 + k < n && coefficient < -precision) {
                            const double ratio = -elem(ob[jb + j], k) / coefficient;
                            if (ratio > -INFINITY && ki_better(-ratio, c0 + k + 1, e.k, e.i)) {
                                e.k = -ratio;
                                e.i = c0 + k + 1;
                            }
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
            col = __builtin_amdgcn_readfirstlane(e.i);
        }
        if constexpr (CHECK) { // :98,137 hasCycle before the pivot: workgroup 0 (it maintains the basis) decides for everybody
            int cycled = 0;
            if (b == 0) { // (the basis is updated by my lane 0 behind my own barriers; agent-scope loads read it at L2)
                const int leaving = __hip_atomic_load(d.var + w + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int entering = __hip_atomic_load(d.var + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cycled = has_cycle(C, hist_len, leaving, entering, &sh_flag) ? 1 : 0;
                if (tid == 0)
                    __hip_atomic_store(d.rc_verdict + par, ((unsigned long long)epoch << 32) | (unsigned)cycled, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
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
            if (cycled) { // ["cycled", NaN]: this pivot is not carried out; the pending ones are, on the way out
                term = YALPS_CYCLED;
                stop = true;
                continue;
            }
        }
        YSTAMP(2); // phase 1: entering column; checkCycles: verdict
        // ---------------- pivot (src/simplex.ts:5-39): it becomes pending pivot number npend ---------------------------
        const int colx = col - 1;
        double *prowN = prow0 + (size_t)npend * pitch, *colvN = colv0 + npend * rpw, *nqvN = nqv0 + npend * rpw;
        // my rows' pivot-column entries as they are now, the quotient, the objective row's entry (LDS: complete since
        // price()'s barrier of the previous pivot)
        if (phase == 2) {
            for (int i = tid; i < my_rows; i += T) colvN[i] = lav[i];
            __syncthreads();
        } else {
            column_now(colx, colvN);
        }
        // the objective row's entry of the pivot column, from the lane that holds it (per-lane predicates over compile-time
        // (j, e): a select chain over a run-time unit index is turned into a dynamically indexed scratch array by hipcc)
        {
            int tc = tid;
            asm volatile("" : "+v"(tc));
#pragma unroll
            for (int j = 0; j < J; j++) {
                if (2 * (tc + j * T) == colx) sh_c0 = ob[j].x;
                if (2 * (tc + j * T) + 1 == colx) sh_c0 = ob[j].y;
            }
        }
        const double q = ld_sc1(src + colx), inv_q = 1.0 / q;
        __syncthreads(); // (sh_c0; nobody writes it again before the barriers of the next pivot)
        const double coef0 = sh_c0;
        YSTAMP(3); // my rows' pivot-column entries, quotient (two barriers)
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
        YSTAMP(4); // RHS entries, what replaces the pivot column
        // one pass over the raw row, JC units per lane at a time: normalised -> my scratch (:14-25; FLUSHED marks what pivot()
        // zeroed), my objective replica updated (:27-38 for row 0), and priced (:71-79) while it is in registers
        const bool touched0 = fabs(coef0) > 1e-16;
        const double nq0 = -coef0 / q; // :36 for the objective row
        const __amdgpu_buffer_rsrc_t rsrc_new = rsrc_of(prowN);
        unsigned nzmask = 0;
        KI best = {INFINITY, INT_MAX};
        int tp = tid;
        asm volatile("" : "+v"(tp)); // (opaque: see price())
#pragma unroll
        for (int jb = 0; jb < J; jb += JC) {
            double2 pv[JC];
#pragma unroll
            for (int j = 0; j < JC; j++) pv[j] = row_ld16<AUX_SC1>(rsrc_src, lane_off + 16 * T * (jb + j), 0);
#pragma unroll
            for (int j = 0; j < JC; j++) {
                const int c0 = 2 * (tp + (jb + j) * T);
                double2 pn, ov = ob[jb + j];
                bool nzk[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const double v = elem(pv[j], k);
                    nzk[k] = fabs(v) > 1e-16;
                    const double vn = nzk[k] ? v / q : 0.0;
                    pn = with_elem(pn, k, nzk[k] ? vn : flushed);
                    if (nzk[k]) nzmask |= 1u << (2 * (jb + j) + k);
                    double o1 = elem(ov, k);
                    if (touched0) {
                        if (c0 + k == colx)
                            o1 = nq0;
                        else if (nzk[k]) {
                            const double prod = coef0 * vn;
                            o1 = o1 - prod;
                        }
                    }
                    ov = with_elem(ov, k, o1);
                    if (c0 + k < n && o1 > precision && ki_better(-o1, c0 + k + 1, best.k, best.i)) {
                        best.k = -o1;
                        best.i = c0 + k + 1;
                    }
                }
                ob[jb + j] = ov;
                row_st16<AUX_PLAIN>(rsrc_new, lane_off + 16 * T * (jb + j), 0, pn);
            }
            __builtin_amdgcn_sched_barrier(0); // (the next JC units' loads stay behind this group's arithmetic: registers)
        }
        {
            const bool fast = __builtin_amdgcn_ballot_w64(((nzmask | padmask) & FULL) != FULL) == 0; // (per wave)
            if ((tid & 63) == 0) sh_fast[npend][tid >> 6] = fast ? 1 : 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (my scratch stores are out before the barrier below: other waves' scalar chains read them at L2)
        YSTAMP(5); // the raw pivot row (sc1) -> normalised -> scratch, objective replica, priced; stores drained
        if (tid == 0) { // (published to the workgroup by price()'s barrier, like prow / rhsv)
            sh_pl[npend] = lslot;
            sh_pc[npend] = colx;
        }
        npend += 1;
        iter += 1.0;
        pivots += 1;
        done += 1;
        best = block_argmin<T>(best, sk, si, slot); // la of the next pivot (its barrier also publishes the scratch row / rhsv / the pending scalars)
        slot ^= 1;
        la = __builtin_amdgcn_readfirstlane(best.i == INT_MAX ? 0 : best.i); // (uniform: kept in a scalar register)
        check();
        YSTAMP(6); // arg-max of the pricing (barriers)
        // The doubles behind column n of a device row are padding (rows are 128 bytes apart): the pass over the pivot row has marked them
        // FLUSHED like any other zero; the select-free path of the sweep multiplies every lane's units by this row, so the
        // lane that holds them overwrites its own marks with a finite 0.0 (same lane, same address: in order).  The checkCycles
        // forms have no register left for that (at 8 and 16 units per lane they would spill): there the one or two waves that
        // hold padding never take the select-free path instead (padmask covers only the units beyond the pitch).
        if constexpr (!CHECK) {
            const int u_first = n >> 1, d_lane = (tid - u_first) & (T - 1); // (T is a power of two)
            if (d_lane < (pitch >> 1) - u_first) {
                const int c0p = 2 * (u_first + d_lane);
                if (c0p >= n) (prow0 + (size_t)(npend - 1) * pitch)[c0p] = 0.0;
                (prow0 + (size_t)(npend - 1) * pitch)[c0p + 1] = 0.0;
            }
        }
        if (!stop) {
            if (phase == 2) column_now(la - 1, lav); // my rows' entries of column la after every pivot so far
            YSTAMP(7); // my rows' entries of the next entering column (scalar chains)
            const KI cand = candidate(phase);        // (la > 0 here: check() stops phase 2 without an entering column)
            if constexpr (TWO) {
                // the workgroups of my XCD read my candidate row from memory if it wins: when this pivot's sweep is due it comes
                // BEFORE my key record (with the key out first, they could read a row I am still sweeping)
                if (npend == depth) flush_pending();
            }
            publish(cand);
            YSTAMP(8); // my candidate (arg-min), its key record (single hand-off: + the row through the pending pivots)
        }
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12 (off the critical path)
            const int leaving = d.var[w + row], entering = d.var[col];
            __hip_atomic_store(d.var + w + row, entering, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.var + col, leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d.pos[leaving] = col;
            d.pos[entering] = w + row;
        }
        // ---------------- the rows: only every depth-th pivot (or on the way out) ----------------------------------------
        YSTAMP(9); // basis bookkeeping (workgroup 0)
        if (npend == depth || stop)
            flush_pending();
        else
            __syncthreads(); // (colv / rhsv / lav of this pivot are complete before the next round's lanes read them)
        YSTAMP(10); // the sweep (every depth-th pivot) / barrier
    }
#ifdef YALPS_STAMPS
    if (tid == 0 && d.dbg) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        unsigned long long *out = d.dbg + (size_t)b * STAMP_WORDS;
#pragma unroll
        for (int k = 0; k < 20; k++) out[k] += st_acc[k];
        out[20] += (unsigned long long)done;
        out[21] += t1 - st_t0;
        out[22] += r1 - st_r0;
    }
#endif
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
