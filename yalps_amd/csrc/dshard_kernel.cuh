// dshard_kernel.cuh -- one pivot of a ROW SHARD with delayed row updates (the launch-per-pivot form of stream3_kernel)
// Part of libyalps_hip.so; included by persistent_dshard.hip inside its unnamed namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// dshard_kernel<T lanes, J units per lane per row, NT>: the MODE_SHARD step of wide_kernel<.., true, ..> (same launch
// protocol: state d.st[parity] -> d.st[parity ^ 1], the all-gathered slots of SURVEY.md 8e in `gather`, this rank's partials
// out) with stream3_kernel's data flow: the pivot just decided is NOT swept over my rows.  It becomes pending pivot number
// npend: its normalised row goes to d.dpend[npend], my rows' entries of its column (as they are NOW: memory + the pivots
// pending before it, a scalar chain per row) and what replaces them (:25, :36) to d.dcolv / d.dnqv[npend], the RHS
// column and the objective replica are updated at once -- everything the next decision reads.  My rows' entries of the
// next entering column come out of the same scalar chains (d.dlav), the partials from them.  Every d.delay_depth pivots,
// and when the solve ends, every touched row of mine is streamed ONCE and gets all pending eliminations in registers,
// each with its own separately rounded multiply and subtract in the reference's order: bit for bit what that many sweeps
// leave (src/simplex.ts:5-39).  The candidate rows a rank sends get the pending pivots applied on their way into the
// all-gather slot (dshard_select_kernel): what travels is the row as the reference would hold it.
// The sweep itself is panel_flush.cuh: the pending rows pass through LDS one column panel at a time (round 3; up to
// DSHARD_MAXD = 16 pending pivots of 1024 columns).
// State between launches lives in global memory (LDS does not survive a launch): d.dstate[parity] {npend, the pending
// pivots' rows and columns}, d.dcolv / d.dnqv [depth][hcap], d.dlav [hcap], d.dpend [depth][pitch].  ONE copy of the
// pending rows serves all XCDs here (stream3_kernel needs one per XCD): every workgroup stores the same bytes, reads back
// within the launch only what it stored itself, and a launch boundary writes the L2s back and invalidates them.
// ------------------------------------------------------------------------------------------
template <int T, int J, bool NT, bool PANEL>
__global__ __launch_bounds__(T) void dshard_kernel(Desc d, int parity, int /*mode: MODE_SHARD*/, int /*force*/, const double *gather) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ int sh_nt;
    constexpr int MAXD = DSHARD_MAXD;
    __shared__ int sh_pl[MAXD], sh_pc[MAXD]; // the pending pivots, oldest first: my slot of the pivot row (-1: not mine), pivot column (mat index)
    constexpr int JC = J > 8 ? 8 : J;        // units per lane that pass through registers at a time (a pivot row being decided)
    constexpr int PU = DSHARD_PANEL_UNITS;   // 16-byte units of a row per panel of the sweep (panel_flush.cuh)
    extern __shared__ __attribute__((aligned(16))) double sm_dyn[]; // colv[depth][rpw], nqv[depth][rpw], lav[rpw], rhsv[rpw], tlist[rpw] (int), panel[depth][2 PU]

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
#ifdef YALPS_STAMPS
    // diagnostic build: stage sums of this launch, added to d.dbg[b] on the way out (stages: tools/shard_stages.py)
    unsigned long long st_acc[20] = {}, st_last = 0, st_t0 = 0, st_r0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0)::"memory");
    st_last = st_t0;
#endif
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const DelayState *Din = d.dstate + parity;
    DelayState *Dout = d.dstate + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING || Sin->pause) {
        if (b == 0 && tid == 0) {
            state_copy(Sout, Sin);
            *Dout = *Din;
        }
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch, hcap = d.hcap;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int mbuf = Sin->mbuf, pbuf = Sin->pbuf, la_in = Sin->la;
    double *mat = d.mat[mbuf];
    double *rhs = d.rhs[mbuf];
    const double *objA = d.obj[pbuf]; // the objective row as every workgroup reads it; workgroup 0 writes the next one
    double *objB = d.obj[pbuf ^ 1];
    const int64_t pivots_in = Sin->pivots, hist_len_in = Sin->hist_len;
    int phase = Sin->phase;
    double iter = Sin->iter;
    bool phase_switched = false;
    int slot = 0;
    const int rpw = (hcap + NB - 1) / NB;
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    const int depth = d.delay_depth < 1 ? 1 : d.delay_depth > MAXD ? MAXD : d.delay_depth;
    double *colv0 = sm_dyn, *nqv0 = colv0 + (size_t)depth * rpw, *lav = nqv0 + (size_t)depth * rpw, *rhsv = lav + rpw;
    int *tlist = reinterpret_cast<int *>(rhsv + rpw), *tmask = tlist + (rpw + 3) / 4 * 4, *tpiv = tmask + (rpw + 3) / 4 * 4;
    double *panel = rhsv + rpw + (rpw + 3) / 4 * 6; // (behind tlist, tmask and tpiv, 16-byte aligned: rpw ints each, rounded up to a multiple of four)
    const double flushed = __longlong_as_double((long long)FLUSHED);
    double *const prow0 = d.dpend;
    int npend = Din->npend;
    npend = npend < 0 ? 0 : npend > depth ? depth : npend; // (`depth` pending: the sweep of its own launch did not run -- swept below, before this pivot)
    const bool lav_valid = Din->lav_valid != 0;

    // basis bookkeeping of the pivot before this one (src/simplex.ts:7-12), as wide_kernel: its loads are this launch's oldest
    const bool swapper = b == 0 && tid == 0 && Sin->swap_valid;
    if (swapper) {
        const int sw_row = Sin->swap_row, sw_col = Sin->swap_col;
        const int leaving = d.var[d.w + sw_row], entering = d.var[sw_col];
        d.var[d.w + sw_row] = entering;
        d.var[sw_col] = leaving;
        d.pos[leaving] = sw_col;
        d.pos[entering] = d.w + sw_row;
    }

    const int lane_off = 16 * tid, row_bytes = pitch * 8;
    auto rsrc_of = [&](const double *row_ptr) __attribute__((always_inline)) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(row_ptr);
        const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double *>(u), 0, row_bytes, 0x00020000);
    };

    // ---- what the earlier launches left: the pending pivots' scalars for my rows, my rows' RHS and look-ahead column ----
    static_assert(MAXD <= T, "one lane per pending pivot");
    if (tid < MAXD) {
        const int lr = tid < npend ? Din->pl[tid] : -1;
        sh_pl[tid] = (lr >= 0 && lr % NB == b) ? lr / NB : -1;
        sh_pc[tid] = tid < npend ? Din->pc[tid] : 0;
    }
    for (int i = tid; i < my_rows; i += T) {
        const int r = b + NB * i;
        for (int p = 0; p < npend; p++) {
            colv0[p * rpw + i] = d.dcolv[(size_t)p * hcap + r];
            nqv0[p * rpw + i] = d.dnqv[(size_t)p * hcap + r];
        }
        rhsv[i] = rhs[r];
        lav[i] = lav_valid ? d.dlav[r] : 0.0;
    }
    __syncthreads();
    YSTAMP(0); // state, the pending pivots' scalars of my rows -> LDS

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
        asm volatile("" : "+v"(t0));
        for (int i = t0; i < my_rows; i += T) {
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
    // every touched row of mine streamed once, all pending eliminations in registers; afterwards nothing is pending
    // (the objective row is not swept: workgroup 0 copies the replica -- `obj_now`, the same arithmetic pivot by pivot -- over it; with it the
    // first workgroup had one row more than the others wherever a rank holds 2^k rows: a fifth trip of 8 waves x 2 rows per panel for ONE row)
    auto flush_pending = [&](const double *obj_now) __attribute__((always_inline)) {
        if (npend == 0) return; // (uniform)
        if (b == 0) {
            const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(obj_now), rs_0 = rsrc_of(mat);
#pragma unroll 1
            for (int jb = 0; jb < J; jb += 4) {
                double2 o[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (jb + j < J) o[j] = row_ld16<AUX_PLAIN>(rs_o, lane_off + 16 * T * (jb + j), 0); // (its lanes' own stores where this launch wrote it: in order)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (jb + j < J) row_st16<NT ? AUX_NT : AUX_PLAIN>(rs_0, lane_off + 16 * T * (jb + j), 0, o[j]);
            }
        }
        if (tid < 64) {         // compact list of my touched rows (wave 0)
            int cnt = 0;
            for (int base = 0; base < my_rows; base += 64) {
                const int i = base + tid;
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
                    const int k = cnt + __popcll(m & ((1ull << tid) - 1ull));
                    tlist[k] = i;
                    tmask[k] = msk;
                    tpiv[k] = pvm;
                }
                cnt += __popcll(m);
            }
            if (tid == 0) sh_nt = cnt;
        }
        __syncthreads();
        // (panel_flush.cuh: the pending rows staged in LDS one column panel at a time, a wave per row, eight units per lane, two rows in flight per wave)
        if constexpr (PANEL)
            panel_flush<T, PU, 64, YALPS_PANEL_D, YALPS_PANEL_SETS, NT, 8, (J < 16)>(mat, pitch, b, NB, prow0, npend, colv0, nqv0, rpw, sh_pl, sh_pc, tlist, tmask, tpiv, sh_nt, panel, rsrc_of YSTAMP_ARGS);
        else // (few rows per workgroup: the pending rows straight from L2, round 2's form)
            direct_flush<T, J, (J > 8 ? 4 : 3), NT>(mat, pitch, b, NB, prow0, npend, colv0, nqv0, rpw, sh_pl, sh_pc, tlist, sh_nt, rsrc_of);
        npend = 0;
    };
    auto write_state = [&](int status, int phase_, int la_, int pbuf_, int swap_valid_, int swap_row_, int swap_col_, int64_t hist_len_,
                           double iter_, double result_, int64_t pivots_) __attribute__((always_inline)) {
        Sout->status = status;
        Sout->phase = phase_;
        Sout->bootstrap = 0;
        Sout->la = la_;
        Sout->pbuf = pbuf_;
        Sout->mbuf = mbuf;
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

    // ---------------- decide: every rank's two candidates arrived with the all-gather (wide_kernel, MODE_SHARD) ----------------
    const int gstride = SHARD_HDR + 2 * pitch, ncand = d.nshards;
    Part p_rhs, p_ratio;
    {
        const double *slot_ = gather + (size_t)(tid < ncand ? tid : 0) * gstride;
        p_ratio.key = slot_[0];
        p_ratio.idx = (int)slot_[1];
        p_rhs.key = slot_[2];
        p_rhs.idx = (int)slot_[3];
    }
    auto owner_slot = [&](int grow) __attribute__((always_inline)) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < MAX_SHARDS; k++)
            if (k < d.nshards && grow >= d.bounds[k]) g = k;
        return gather + (size_t)g * gstride;
    };
    const __amdgpu_buffer_rsrc_t rs_objA = rsrc_of(objA);
    int row = 0, col = 0, term = RUNNING;
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
            if (c.i == INT_MAX) { // :120
                phase = 2;
                iter = 0.0;
                phase_switched = true;
                continue;
            }
            row = c.i;
            const __amdgpu_buffer_rsrc_t rs1 = rsrc_of(owner_slot(row) + SHARD_HDR + pitch);
            KI e = {INFINITY, INT_MAX};
#pragma unroll 1
            for (int jb = 0; jb < J; jb += JC) { // :123-134
                double2 cr[JC], ob[JC];
#pragma unroll
                for (int j = 0; j < JC; j++) {
                    cr[j] = row_ld16<AUX_PLAIN>(rs1, lane_off + 16 * T * (jb + j), 0);
                    ob[j] = row_ld16<AUX_PLAIN>(rs_objA, lane_off + 16 * T * (jb + j), 0);
                }
#pragma unroll
                for (int j = 0; j < JC; j++) {
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int cc = 2 * (tid + (jb + j) * T) + k;
                        const double coefficient = elem(cr[j], k);
                        if (cc < n && coefficient < -precision) {
                            const double ratio = -elem(ob[j], k) / coefficient;
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
            if (e.i == INT_MAX) { // :135
                term = YALPS_INFEASIBLE;
                break;
            }
            col = e.i;
            break;
        } else {
            col = la_in;
            if (col == 0) { // :80
                term = YALPS_OPTIMAL;
                term_result = round_to_precision(rhs[0], precision);
                break;
            }
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
    const int check = C->check_cycles ? 1 : 0;
    if (term == RUNNING && check && d.cyc_verdict[parity & 1]) term = YALPS_CYCLED; // :98,137: shard_cycle_kernel's verdict on this pivot
    YSTAMP(1); // decide (the gathered records, phase 1: the entering column)
    if (term == RUNNING && npend >= depth) term = SHARD_SWEEP_MISSING; // (a launch needs room for one more pending pivot: d.ext_sweep's launch did not run -- the host fails on this status)
    if (term != RUNNING) { // the solve ends here: the pending pivots are carried out on the way out
        flush_pending(objA);
        if (b == 0 && tid == 0) {
            write_state(term, phase, la_in, pbuf, 0, 0, 0, (phase_switched ? 0 : hist_len_in) + (term == YALPS_CYCLED ? check : 0), iter, term_result, pivots_in);
            DelayState z = {};
            *Dout = z;
        }
        return;
    }

    // ---------------- pivot (src/simplex.ts:5-39): it becomes pending pivot number npend ---------------------------
    const int colx = col - 1;
    const double *gslot = owner_slot(row);
    const double *mrow = gslot + SHARD_HDR + (phase == 1 ? pitch : 0); // the pivot row as its owner sent it: every earlier pivot applied
    const int lrow = (row >= d.bounds[d.shard_rank] && row < d.bounds[d.shard_rank + 1]) ? row - d.row_base : -1;
    const int lslot = (lrow >= 0 && lrow % NB == b) ? lrow / NB : -1;
    const double q = mrow[colx], coef0 = objA[colx], rhs_row = gslot[phase == 1 ? 5 : 4], inv_q = 1.0 / q;
    double *colvN = colv0 + npend * rpw, *nqvN = nqv0 + npend * rpw;
    if (phase == 2 && lav_valid) { // the look-ahead of the launch before this one priced exactly this column
        for (int i = tid; i < my_rows; i += T) colvN[i] = lav[i];
        __syncthreads();
    } else {
        column_now(colx, colvN);
    }
    YSTAMP(2); // my rows' entries of the pivot column
    const bool nz_rhs = fabs(rhs_row) > 1e-16;
    const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
    for (int i = tid; i < my_rows; i += T) { // RHS entries of my rows (:33 at column 0)
        const int r = b + NB * i;
        const double coef = colvN[i];
        if (i == lslot)
            rhsv[i] = pn_rhs;
        else if (fabs(coef) > 1e-16 && nz_rhs) {
            const double prod = coef * pn_rhs;
            rhsv[i] = rhsv[i] - prod;
        }
        const double nq = i == lslot ? inv_q : -coef / q; // what replaces the pivot column (:25, :36)
        nqvN[i] = nq;
        rhs[r] = rhsv[i];
        d.dcolv[(size_t)npend * hcap + r] = coef;
        d.dnqv[(size_t)npend * hcap + r] = nq;
    }
    YSTAMP(3); // RHS, what replaces the pivot column; the pending scalars stored
    // one pass over the pivot row, JC units per lane at a time: normalised -> d.dpend[npend] (:14-25; FLUSHED marks what pivot()
    // zeroed), the objective replica of the next launch (:27-38 for row 0; workgroup 0 stores it), priced (:71-79) in registers
    const bool touched0 = fabs(coef0) > 1e-16;
    const double nq0 = -coef0 / q; // :36 for the objective row
    const __amdgpu_buffer_rsrc_t rsrc_src = rsrc_of(mrow), rsrc_new = rsrc_of(prow0 + (size_t)npend * pitch), rs_objB = rsrc_of(objB);
    KI best = {INFINITY, INT_MAX};
#pragma unroll 1
    for (int jb = 0; jb < J; jb += JC) {
        double2 pv[JC], ob[JC];
#pragma unroll
        for (int j = 0; j < JC; j++) {
            pv[j] = row_ld16<AUX_PLAIN>(rsrc_src, lane_off + 16 * T * (jb + j), 0);
            ob[j] = row_ld16<AUX_PLAIN>(rs_objA, lane_off + 16 * T * (jb + j), 0);
        }
#pragma unroll
        for (int j = 0; j < JC; j++) {
            const int c0 = 2 * (tid + (jb + j) * T);
            double2 pn, ov = ob[j];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double v = elem(pv[j], k);
                const bool nzk = fabs(v) > 1e-16;
                const double vn = nzk ? v / q : 0.0;
                pn = with_elem(pn, k, nzk ? vn : flushed);
                double o1 = elem(ov, k);
                if (touched0) {
                    if (c0 + k == colx)
                        o1 = nq0;
                    else if (nzk) {
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
            if (b == 0) row_st16<AUX_PLAIN>(rs_objB, lane_off + 16 * T * (jb + j), 0, ov);
            row_st16<AUX_PLAIN>(rsrc_new, lane_off + 16 * T * (jb + j), 0, pn);
        }
    }
    // The doubles behind column n of a device row are padding (rows are 128 bytes apart): the pass above has marked them
    // FLUSHED like any other zero; the select-free path of the sweep multiplies every lane's units by this row, so the
    // lane that holds them overwrites its own marks with a finite 0.0 (same lane, same address: in order).
    {
        const int u_first = n >> 1, d_lane = (tid - u_first) & (T - 1); // (T is a power of two)
        if (d_lane < (pitch >> 1) - u_first) {
            const int c0p = 2 * (u_first + d_lane);
            if (c0p >= n) (prow0 + (size_t)npend * pitch)[c0p] = 0.0;
            (prow0 + (size_t)npend * pitch)[c0p + 1] = 0.0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (my stores of the pending row are out before the barrier below: the scalar chains read them at L2)
    YSTAMP(4); // the pivot row: normalised + stored, objective replica, priced in registers
    if (tid == 0) {
        sh_pl[npend] = lslot;
        sh_pc[npend] = colx;
    }
    npend += 1;
    best = block_argmin<T>(best, sk, si, slot); // (its barrier also publishes the pending row / rhsv / the pending scalars to my waves)
    slot ^= 1;
    const int la = best.i == INT_MAX ? 0 : best.i;
    YSTAMP(5); // arg-max of the pricing (barriers)

    // ---------------- my candidates for the next pivot: from scalars ----------------------------------------------------
    if (la > 0)
        column_now(la - 1, lav); // my rows' entries of column la after every pivot so far
    else
        __syncthreads();
    YSTAMP(6); // my rows' entries of the next entering column (scalar chains)
    KI cand_ratio = {INFINITY, INT_MAX}, cand_rhs = {INFINITY, INT_MAX};
    for (int i = tid; i < my_rows; i += T) {
        const int r = b + NB * i;
        if (la > 0) d.dlav[r] = lav[i];
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
    YSTAMP(7); // my candidates, two arg-mins, the partials
    // ---------------- the rows: only every depth-th pivot ------------------------------------------------------------------
    const bool ext = d.ext_sweep != 0; // the sweep is a launch of its own behind this one (dsweep_kernel.cuh): the pivots stay pending
    const int npend_out = (npend == depth && !ext) ? 0 : npend;
    if (b == 0 && tid == 0) {
        write_state(RUNNING, phase, la, pbuf ^ 1, 1, row, col, (phase_switched ? 0 : hist_len_in) + check, iter + 1.0, NAN, pivots_in + 1);
        DelayState o = {};
        o.npend = npend_out;
        o.lav_valid = la > 0 ? 1 : 0;
        for (int p = 0; p < MAXD; p++) {
            o.pl[p] = p < npend_out ? (p == npend - 1 ? lrow : Din->pl[p]) : -1;
            o.pc[p] = p < npend_out ? (p == npend - 1 ? colx : Din->pc[p]) : 0;
        }
        *Dout = o;
    }
    YSTAMP(8); // state
    if (npend == depth && !ext) flush_pending(objB);
    YSTAMP(9); // the sweep (every depth-th launch)
#ifdef YALPS_STAMPS
    if (tid == 0 && d.dbg) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        unsigned long long *out = d.dbg + (size_t)b * STAMP_WORDS;
#pragma unroll
        for (int k = 0; k < 20; k++) out[k] += st_acc[k];
        out[20] += 1ull;
        out[21] += t1 - st_t0;
        out[22] += r1 - st_r0;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// dshard_select_kernel: shard_select_kernel (this rank's two candidates + their rows into its all-gather slot) for a shard
// with delayed row updates: the rows leave with the pending pivots applied, element by element, each with its own rounding
// (the same arithmetic as the sweep: :14-25 for a row that was a pivot row, :31-36 otherwise).
// ------------------------------------------------------------------------------------------
// ST lanes per workgroup: 1024 (up to 1024 partials), or 256 where the shard has at most 256 workgroups -- four times the
// workgroups for the same units: a lane's chain is (1 + npend) dependent-free loads of ONE unit, and what bounds the kernel is
// how many CUs pull the pending rows through their L2 ports (2049 x 16385: 46.9 -> 46.4 us per pivot of the whole loop).
template <int ST>
__global__ __launch_bounds__(ST) void dshard_select_kernel(Desc d, int parity, double *send) {
    constexpr int MAXD = DSHARD_MAXD;
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_coef[2][MAXD], sh_patch[2][MAXD];
    __shared__ int sh_piv[2][MAXD], sh_col[MAXD];
    const YState *S = d.st + parity;
    const DelayState *D = d.dstate + parity;
    const int tid = threadIdx.x, NB = d.nb, pitch = d.pitch, hcap = d.hcap;
    const bool idle = S->status != RUNNING || S->pause || S->bootstrap;
    KI cr = {INFINITY, INT_MAX}, cn = {INFINITY, INT_MAX};
    if (!idle && tid < NB) {
        const Part a = d.part_ratio[S->pbuf][tid], c = d.part_rhs[S->pbuf][tid];
        cr.k = a.key;
        cr.i = a.idx;
        cn.k = c.key;
        cn.i = c.idx;
    }
    cr = block_argmin<ST>(cr, sk, si, 0);
    cn = block_argmin<ST>(cn, sk, si, 1);
    const double *mat = d.mat[S->mbuf], *rhs = d.rhs[S->mbuf];
    const int lr = cr.i == INT_MAX ? 0 : cr.i - d.row_base, ln = cn.i == INT_MAX ? 0 : cn.i - d.row_base;
    int npend = idle ? 0 : D->npend;
    npend = npend < 0 ? 0 : npend > MAXD ? MAXD : npend;
    const int units = pitch / 2;
    const double2 *r0 = reinterpret_cast<const double2 *>(mat + (size_t)lr * pitch), *r1 = reinterpret_cast<const double2 *>(mat + (size_t)ln * pitch);
    double2 *o0 = reinterpret_cast<double2 *>(send + SHARD_HDR), *o1 = reinterpret_cast<double2 *>(send + SHARD_HDR + pitch);
    // a lane's unit of a candidate row and of every pending row: issued BEFORE the candidates' scalars are fetched and published to the
    // workgroup (neither depends on the other; behind the barrier they were one more memory round trip on every pivot's chain)
    auto fetch = [&](int u, double2 &v, double2 (&pn)[MAXD]) __attribute__((always_inline)) {
        const int which = u < units ? 0 : 1, uu = which ? u - units : u;
        v = which ? r1[uu] : r0[uu];
#pragma unroll
        for (int p = 0; p < MAXD; p++)
            pn[p] = p < npend ? reinterpret_cast<const double2 *>(d.dpend + (size_t)p * pitch)[uu] : make_double2(0.0, 0.0);
    };
    const int ufirst = blockIdx.x * ST + tid, ustride = gridDim.x * ST;
    double2 v_first = make_double2(0.0, 0.0), pn_first[MAXD];
    if (ufirst < 2 * units) fetch(ufirst, v_first, pn_first);
    if (tid < 2 * MAXD) {
        const int which = tid / MAXD, p = tid % MAXD, r = which ? ln : lr;
        if (p < npend) {
            sh_coef[which][p] = d.dcolv[(size_t)p * hcap + r];
            sh_patch[which][p] = d.dnqv[(size_t)p * hcap + r];
            sh_piv[which][p] = D->pl[p] == r ? 1 : 0;
            if (which == 0) sh_col[p] = D->pc[p];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        send[0] = cr.k;
        send[1] = (double)cr.i;
        send[2] = cn.k;
        send[3] = (double)cn.i;
        send[4] = rhs[lr];
        send[5] = rhs[ln];
        send[6] = 0.0;
        send[7] = 0.0;
    }
    auto finish = [&](int u, double2 v, const double2 (&pn)[MAXD]) __attribute__((always_inline)) {
        const int which = u < units ? 0 : 1, uu = which ? u - units : u;
#pragma unroll
        for (int p = 0; p < MAXD; p++) {
            if (p >= npend) continue;
            const double coef = sh_coef[which][p];
            const bool piv = sh_piv[which][p] != 0;
            if (!(piv || fabs(coef) > 1e-16)) continue; // :31
            const bool f0 = (unsigned long long)__double_as_longlong(pn[p].x) != FLUSHED;
            const bool f1 = (unsigned long long)__double_as_longlong(pn[p].y) != FLUSHED;
            if (piv) {
                v.x = f0 ? pn[p].x : 0.0;
                v.y = f1 ? pn[p].y : 0.0;
            } else {
                const double px = coef * pn[p].x, py = coef * pn[p].y;
                const double nx = v.x - px, ny = v.y - py;
                v.x = f0 ? nx : v.x;
                v.y = f1 ? ny : v.y;
            }
            const int colxp = sh_col[p];
            if (2 * uu == (colxp & ~1)) {
                if (colxp & 1)
                    v.y = sh_patch[which][p];
                else
                    v.x = sh_patch[which][p];
            }
        }
        if (which)
            o1[uu] = v;
        else
            o0[uu] = v;
    };
    if (ufirst < 2 * units) finish(ufirst, v_first, pn_first);
    for (int u = ufirst + ustride; u < 2 * units; u += ustride) {
        double2 v, pn[MAXD];
        fetch(u, v, pn);
        finish(u, v, pn);
    }
}
