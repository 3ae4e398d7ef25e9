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
