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
