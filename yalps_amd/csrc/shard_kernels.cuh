// shard_kernels.cuh -- per-rank select kernel of the row-sharded solve; basis-swap flush
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// shard_select_kernel: one workgroup per rank, between two pivots of a row-sharded solve.
// Reduces this rank's per-workgroup partials to its two candidates and packs them, WITH the
// candidate rows, into the rank's slot of the all-gather (so the selection and the pivot-row
// broadcast of SURVEY.md 8e are a single collective of nshards x (8 + 2*pitch) doubles):
//   [0] ratio key  [1] ratio row (global)  [2] rhs key  [3] rhs row (global)
//   [4] RHS entry of the ratio row  [5] RHS entry of the rhs row  [6..7] pad
//   [8 .. 8+pitch) raw ratio-candidate row   [8+pitch .. 8+2*pitch) raw rhs-candidate row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void shard_select_kernel(Desc d, int parity, double *send) {
    // Grid of gridDim.x workgroups: every one reduces the (<= 1024) partials for itself -- identical inputs, identical
    // result -- and copies its slice of the two candidate rows, 16 bytes per lane (was: one workgroup, 8 bytes per lane:
    // 2 x 131 KB at w = 16385 took longer than the all-gather it feeds).
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    const YState *S = d.st + parity;
    const int tid = threadIdx.x, NB = d.nb, pitch = d.pitch;
    const bool idle = S->status != RUNNING || S->pause || S->bootstrap;
    KI cr = {INFINITY, INT_MAX}, cn = {INFINITY, INT_MAX};
    if (!idle && tid < NB) {
        const Part a = d.part_ratio[S->pbuf][tid], c = d.part_rhs[S->pbuf][tid];
        cr.k = a.key;
        cr.i = a.idx;
        cn.k = c.key;
        cn.i = c.idx;
    }
    cr = block_argmin<1024>(cr, sk, si, 0);
    cn = block_argmin<1024>(cn, sk, si, 1);
    const double *mat = d.mat[S->mbuf], *rhs = d.rhs[S->mbuf];
    const int lr = cr.i == INT_MAX ? 0 : cr.i - d.row_base, ln = cn.i == INT_MAX ? 0 : cn.i - d.row_base;
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
    const int units = pitch / 2; // (pitch is a multiple of 16 doubles, SHARD_HDR of 2: every access below is 16-byte aligned)
    const double2 *r0 = reinterpret_cast<const double2 *>(mat + (size_t)lr * pitch), *r1 = reinterpret_cast<const double2 *>(mat + (size_t)ln * pitch);
    double2 *o0 = reinterpret_cast<double2 *>(send + SHARD_HDR), *o1 = reinterpret_cast<double2 *>(send + SHARD_HDR + pitch);
    for (int u = blockIdx.x * 1024 + tid; u < 2 * units; u += gridDim.x * 1024) {
        if (u < units)
            o0[u] = r0[u];
        else
            o1[u - units] = r1[u - units];
    }
}

// ------------------------------------------------------------------------------------------
// shard_cycle_kernel: hasCycle (src/simplex.ts:44-63, called at :98 and :137) for a row-sharded solve; ONE workgroup,
// enqueued between the all-gather and the step launch of the same parity when checkCycles is on.
// The step launch is a grid in which every workgroup takes the decision for itself; the detector needs the basis as it
// is (one array, updated by workgroup 0 of the NEXT launch) and appends to ONE history, so it cannot run inside that
// grid without an exchange between its workgroups.  Instead this kernel takes the same decision from the same bytes (the
// gathered records, the objective row, the state: the decide section of pivot_kernel / wide_kernel / dshard_kernel in
// MODE_SHARD), looks the leaving and entering variable up in the basis -- the swap of the pivot before is still pending in
// the state (the step launch applies it) and is applied on the fly --, records the pair, runs the detector with one lane
// per candidate cycle length, and leaves its verdict in d.cyc_verdict[parity]; the step launch reads it after its own
// decision and returns ["cycled", NaN] instead of pivoting (dshard_kernel: after carrying out what is pending).
// Permutations and history are replicated: every rank reaches the same verdict without communication.
// `inplace`: where the step kernel reads the objective row (d.obj[pbuf] for shards swept in place, else row 0 of mat[mbuf]).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void shard_cycle_kernel(Desc d, int parity, const double *gather, int inplace) {
    constexpr int T = 1024;
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ int cyc_flag;
    const YState *S = d.st + parity;
    const YConst *C = d.cst;
    const int tid = threadIdx.x;
    int32_t *verdict = d.cyc_verdict + (parity & 1);
    if (S->status != RUNNING || S->pause || S->bootstrap || !C->check_cycles) {
        if (tid == 0) *verdict = 0;
        return;
    }
    const int n = d.n, pitch = d.pitch;
    const double precision = C->precision, max_pivots = C->max_pivots;
    int phase = S->phase;
    double iter = S->iter;
    bool phase_switched = false;
    int slot = 0;
    const int gstride = SHARD_HDR + 2 * pitch, ncand = d.nshards;
    KI c_ratio = {INFINITY, INT_MAX}, c_rhs = {INFINITY, INT_MAX};
    if (tid < ncand) {
        const double *slot_ = gather + (size_t)tid * gstride;
        c_ratio.k = slot_[0];
        c_ratio.i = (int)slot_[1];
        c_rhs.k = slot_[2];
        c_rhs.i = (int)slot_[3];
    }
    const double *obj = inplace ? d.obj[S->pbuf] : d.mat[S->mbuf];
    int row = 0, col = 0;
    bool have_pivot = false;
    for (;;) {
        if (!(iter < max_pivots)) break; // :69,109
        if (phase == 1) {
            const KI c = block_argmin<T>(c_rhs, sk, si, slot); // :111-119
            slot ^= 1;
            if (c.i == INT_MAX) { // :120
                phase = 2;
                iter = 0.0;
                phase_switched = true;
                continue;
            }
            row = c.i;
            int g = 0;
#pragma unroll
            for (int k = 1; k < MAX_SHARDS; k++)
                if (k < d.nshards && row >= d.bounds[k]) g = k;
            const double *mrow = gather + (size_t)g * gstride + SHARD_HDR + pitch;
            KI e = {INFINITY, INT_MAX};
            for (int cc = tid; cc < n; cc += T) { // :123-134
                const double coefficient = mrow[cc];
                if (coefficient < -precision) {
                    const double ratio = -obj[cc] / coefficient;
                    if (ratio > -INFINITY && ki_better(-ratio, cc + 1, e.k, e.i)) {
                        e.k = -ratio;
                        e.i = cc + 1;
                    }
                }
            }
            e = block_argmin<T>(e, sk, si, slot);
            slot ^= 1;
            if (e.i == INT_MAX) break; // :135 infeasible
            col = e.i;
            have_pivot = true;
            break;
        } else {
            col = S->la;
            if (col == 0) break; // :80 optimal
            const KI c = block_argmin<T>(c_ratio, sk, si, slot); // :83-95
            slot ^= 1;
            if (c.i == INT_MAX) break; // :96 unbounded
            row = c.i;
            have_pivot = true;
            break;
        }
    }
    if (!have_pivot) { // the step launch ends the solve (or the phase) without a pivot
        if (tid == 0) *verdict = 0;
        return;
    }
    // the basis as it is now: memory + the swap still pending in the state (src/simplex.ts:7-12 of the pivot before)
    const int w = d.w;
    auto var_now = [&](int x) __attribute__((always_inline)) {
        if (S->swap_valid) {
            const int a = w + S->swap_row, bcol = S->swap_col;
            if (x == a) return d.var[bcol];
            if (x == bcol) return d.var[a];
        }
        return d.var[x];
    };
    const int leaving = var_now(w + row), entering = var_now(col);
    const int64_t hist_len = phase_switched ? 0 : S->hist_len;
    const bool cyc = has_cycle(C, hist_len, leaving, entering, &cyc_flag);
    if (tid == 0) *verdict = cyc ? 1 : 0;
}

// Applies a pending basis swap left by the last APPLY launch (single-pivot API).
__global__ void flush_swap_kernel(Desc d, int parity) {
    YState *S = d.st + parity;
    if (threadIdx.x == 0 && blockIdx.x == 0 && S->swap_valid) {
        const int w = d.w, row = S->swap_row, col = S->swap_col;
        const int leaving = d.var[w + row], entering = d.var[col];
        d.var[w + row] = entering;
        d.var[col] = leaving;
        d.pos[leaving] = col;
        d.pos[entering] = w + row;
        S->swap_valid = 0;
    }
}
