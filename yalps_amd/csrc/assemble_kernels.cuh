// assemble_kernels.cuh -- on-device assembly of an initial tableau from its non-zero entries
// (SURVEY.md 8f row N2).  The reference's tableauModel (src/tableau.ts:87-134) allocates a zeroed
// dense Float64Array, writes one cell per (variable, constraint) coefficient / RHS / binary row and
// sets both permutations to the identity (:95-98); here the host ships only the written cells
// (16 B each) and the dense rows are produced in HBM.
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// Zero rows [0, height) of tableau buffer 0 (all `pitch` columns, so the padding stays +0.0) and
// write the identity permutations (src/tableau.ts:95-98).
__global__ __launch_bounds__(256) void assemble_clear_kernel(Desc d, int height) {
    const size_t total = (size_t)height * d.pitch, stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 *m2 = reinterpret_cast<double2 *>(d.mat[0]); // pitch is a multiple of 16 doubles
    for (size_t i = i0; i < total / 2; i += stride) m2[i] = make_double2(0.0, 0.0);
    for (size_t i = i0; i < (size_t)height; i += stride) d.rhs[0][i] = 0.0;
    for (size_t i = i0; i < (size_t)(d.w + height); i += stride) {
        d.pos[i] = (int)i;
        d.var[i] = (int)i;
    }
}

// One cell per thread: tableau (row, col) = val, column 0 being the RHS column.  The host has
// checked 0 <= row < height, 0 <= col < w and that no cell occurs twice.
__global__ __launch_bounds__(256) void assemble_scatter_kernel(Desc d, int nnz, const int32_t *__restrict__ row,
                                                               const int32_t *__restrict__ col,
                                                               const double *__restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int r = row[i], c = col[i];
    if (c == 0)
        d.rhs[0][r] = val[i];
    else
        d.mat[0][(size_t)r * d.pitch + (c - 1)] = val[i];
}

// applyCuts (src/branchAndCut.ts:22-61) on the device: node tableau = root optimal tableau (already copied into
// rows [0, h0) of `dst`) + one row per cut (sign, variable, value); one workgroup per cut.  Reads the root's
// rows and basis from the root's own buffers.
__global__ __launch_bounds__(256) void apply_cuts_kernel(Desc dst, const double *__restrict__ root_mat,
                                                         const double *__restrict__ root_rhs,
                                                         const int32_t *__restrict__ root_pos, int h0, int ncuts,
                                                         const int32_t *__restrict__ cut_sign,
                                                         const int32_t *__restrict__ cut_var,
                                                         const double *__restrict__ cut_val) {
    const int i = blockIdx.x, tid = threadIdx.x, w = dst.w, n = dst.n, pitch = dst.pitch;
    if (i < ncuts) {
        const double sign = (double)cut_sign[i], value = cut_val[i];
        const int p = root_pos[cut_var[i]];
        double *row = dst.mat[0] + (size_t)(h0 + i) * pitch;
        if (p < w) { // non-basic at the root: sign * x <= sign * value   (:32-35)
            for (int c = tid; c < pitch; c += 256) row[c] = (c == p - 1) ? sign : 0.0;
            if (tid == 0) dst.rhs[0][h0 + i] = sign * value;
        } else { // basic in root row p - w: substitute that row   (:36-42)
            const double *src = root_mat + (size_t)(p - w) * pitch;
            for (int c = tid; c < pitch; c += 256) row[c] = c < n ? -sign * src[c] : 0.0;
            if (tid == 0) dst.rhs[0][h0 + i] = sign * (value - root_rhs[p - w]);
        }
    }
    // :46-52 the new rows' slack variables extend both permutations with the identity (workgroup 0)
    if (i == 0)
        for (int k = w + h0 + tid; k < w + h0 + ncuts; k += 256) {
            dst.pos[k] = k;
            dst.var[k] = k;
        }
}

// One branch-and-cut node's whole preparation in ONE launch (what used to be three root -> node copies, the cuts' upload,
// apply_cuts_kernel, the state block's upload and the memset of the hand-off words: seven stream operations, each a kernel
// of its own in the runtime with ~4.5 us between dependent ones): workgroups [0, ncuts) build the cut rows exactly as
// apply_cuts_kernel does, reading the cuts straight from pinned host memory (`stage`: [st0 | st1 | cst] then value[ncuts] |
// sign[ncuts] | variable[ncuts]); every workgroup then takes its share of the copies (root rows [0, h0), RHS, both
// permutations), zeroes the hand-off words and moves the state block into place.  Reads root buffers only, writes dst only.
__global__ __launch_bounds__(256) void node_prepare_kernel(Desc dst, const double *__restrict__ root_mat,
                                                           const double *__restrict__ root_rhs,
                                                           const int32_t *__restrict__ root_pos,
                                                           const int32_t *__restrict__ root_var, int h0, int ncuts,
                                                           const char *__restrict__ stage, int state_bytes,
                                                           unsigned long long *__restrict__ sync, long long sync_words) {
    const int b = blockIdx.x, tid = threadIdx.x, w = dst.w, n = dst.n, pitch = dst.pitch;
    if (b < ncuts) {
        const double *cut_val = reinterpret_cast<const double *>(stage + state_bytes);
        const int32_t *cut_sign = reinterpret_cast<const int32_t *>(cut_val + ncuts), *cut_var = cut_sign + ncuts;
        const double sign = (double)cut_sign[b], value = cut_val[b];
        const int p = root_pos[cut_var[b]];
        double *row = dst.mat[0] + (size_t)(h0 + b) * pitch;
        if (p < w) { // non-basic at the root: sign * x <= sign * value   (src/branchAndCut.ts:32-35)
            for (int c = tid; c < pitch; c += 256) row[c] = (c == p - 1) ? sign : 0.0;
            if (tid == 0) dst.rhs[0][h0 + b] = sign * value;
        } else { // basic in root row p - w: substitute that row   (:36-42)
            const double *src = root_mat + (size_t)(p - w) * pitch;
            for (int c = tid; c < pitch; c += 256) row[c] = c < n ? -sign * src[c] : 0.0;
            if (tid == 0) dst.rhs[0][h0 + b] = sign * (value - root_rhs[p - w]);
        }
    }
    const long long gid = (long long)b * 256 + tid, gsz = (long long)gridDim.x * 256;
    { // rows [0, h0): pitch is a multiple of 16 doubles, the rows are contiguous
        const int4 *src = reinterpret_cast<const int4 *>(root_mat);
        int4 *out = reinterpret_cast<int4 *>(dst.mat[0]);
        const long long words = (long long)pitch * h0 / 2;
        for (long long i = gid; i < words; i += gsz) out[i] = src[i];
    }
    for (long long i = gid; i < h0; i += gsz) dst.rhs[0][i] = root_rhs[i];
    for (long long i = gid; i < w + h0; i += gsz) { // :46-52: the old entries, then the new rows' slacks (identity)
        dst.pos[i] = root_pos[i];
        dst.var[i] = root_var[i];
    }
    for (long long i = gid; i < ncuts; i += gsz) {
        dst.pos[w + h0 + i] = (int32_t)(w + h0 + i);
        dst.var[w + h0 + i] = (int32_t)(w + h0 + i);
    }
    for (long long i = gid; i < sync_words; i += gsz) sync[i] = 0ull;
    if (b == gridDim.x - 1) {
        const int4 *src = reinterpret_cast<const int4 *>(stage);
        int4 *out = reinterpret_cast<int4 *>(dst.st);
        for (int i = tid; i < state_bytes / 16; i += 256) out[i] = src[i];
    }
}

// ... and what comes back, in ONE launch instead of three device -> host copies: [error word | st0 | st1] into `ctl_out`,
// column 0 of the buffer the final state names and both permutations into `out` (pinned host memory, written over PCIe;
// layout of yalps_tableau_download_solution's staging: col0[h] | pos[perm_len] ... var at perm_cap).
__global__ __launch_bounds__(256) void node_finish_kernel(Desc d, int h, int perm_len, int perm_cap, int ctl_bytes,
                                                          char *__restrict__ ctl_out, char *__restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    if (blockIdx.x == 0) {
        const int4 *src = reinterpret_cast<const int4 *>(d.rc_err);
        int4 *dstw = reinterpret_cast<int4 *>(ctl_out);
        for (int i = threadIdx.x; i < ctl_bytes / 16; i += 256) dstw[i] = src[i];
    }
    const int mbuf = d.st[1].mbuf & 1;
    const double *rhs = d.rhs[mbuf];
    double *col0 = reinterpret_cast<double *>(out);
    for (int i = gid; i < h; i += gsz) col0[i] = rhs[i];
    int32_t *pos = reinterpret_cast<int32_t *>(out + sizeof(double) * (size_t)h), *var = pos + perm_cap;
    for (int i = gid; i < perm_len; i += gsz) {
        pos[i] = d.pos[i];
        var[i] = d.var[i];
    }
}

