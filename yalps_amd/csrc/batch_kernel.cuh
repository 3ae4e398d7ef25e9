// batch_kernel.cuh -- batched branch-and-cut nodes, one workgroup per node
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once

// ------------------------------------------------------------------------------------------
// batch_kernel: many independent branch-and-cut nodes at once, ONE workgroup per node
// (BASELINE config 4).  A node's LP is the root's optimal tableau plus one
// row per cut (src/branchAndCut.ts:22-61 `applyCuts`), re-solved with simplex() (:127).  The
// root stays resident in HBM; every workgroup builds its node's tableau in its own workspace and
// runs the whole two-phase loop there with workgroup-local synchronisation only -- nodes are
// independent, there is no cross-workgroup communication and no host round trip.
// The loop itself is wg_simplex (wg_simplex.cuh).  LDS = true: the node's tableau is built and
// solved in the LDS of the workgroup's CU and only written to the workspace at the end (nodes of
// small MILPs); LDS = false: it lives in the HBM workspace (L2-resident while it is worked on).
// ------------------------------------------------------------------------------------------

struct BatchDesc {
    const double *root_mat, *root_rhs; // [h0][pitch], [h0]
    const int32_t *root_pos, *root_var; // [w + h0]
    double *ws_mat, *ws_rhs;           // per node: [hmax][pitch], [hmax]
    int32_t *ws_pos, *ws_var;          // per node: [permmax]
    const int32_t *cut_off, *cut_sign, *cut_var; // cuts of node i: [cut_off[i], cut_off[i+1])
    const double *cut_val;
    int32_t *status, *height;
    double *result;
    long long *pivots;
    int32_t w, n, pitch, h0, hmax, permmax;
    double precision, max_pivots;
};

template <int T, bool LDS>
__global__ __launch_bounds__(T) void batch_kernel(BatchDesc d) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double sh_dyn[];

    const int tid = threadIdx.x, node = blockIdx.x;
    const int w = d.w, n = d.n, pitch = d.pitch, h0 = d.h0;
    double *ws_mat = d.ws_mat + (size_t)node * d.hmax * pitch;
    double *ws_rhs = d.ws_rhs + (size_t)node * d.hmax;
    int32_t *ws_pos = d.ws_pos + (size_t)node * d.permmax;
    int32_t *ws_var = d.ws_var + (size_t)node * d.permmax;
    const int c_lo = d.cut_off[node], ncuts = d.cut_off[node + 1] - c_lo;
    const int h = h0 + ncuts;
    // where the node's tableau lives while it is solved
    const int lp = LDS ? small_lds_pitch(n) : pitch, pcols = LDS ? small_pcols(n) : pitch;
    double *mat = LDS ? sh_dyn : ws_mat;
    double *rhs = LDS ? mat + (size_t)d.hmax * lp : ws_rhs;
    double *colbuf = LDS ? rhs + d.hmax : sh_dyn;
    double *prow = colbuf + d.hmax;
    int32_t *pos = LDS ? reinterpret_cast<int32_t *>(prow + lp) : ws_pos;
    int32_t *var = LDS ? pos + ((d.permmax + 1) & ~1) : ws_var;
    const int Uc = wg_unit_lanes(pcols, T), cu0 = tid % Uc, cg0 = tid / Uc, CG = T / Uc;

    // ---- applyCuts (src/branchAndCut.ts:22-61) ----
    for (int r = cg0; r < h0; r += CG) {
        const double *src = d.root_mat + (size_t)r * pitch; // (root padding columns are zero)
        for (int c = cu0; c < pcols; c += Uc) mat[(size_t)r * lp + c] = src[c];
    }
    for (int r = tid; r < h0; r += T) rhs[r] = d.root_rhs[r];
    for (int i = 0; i < ncuts; i++) {
        const double sign = (double)d.cut_sign[c_lo + i], value = d.cut_val[c_lo + i];
        const int p = d.root_pos[d.cut_var[c_lo + i]];
        double *dst = mat + (size_t)(h0 + i) * lp;
        if (p < w) { // non-basic at the root: sign * x <= sign * value   (:32-35)
            for (int c = tid; c < pcols; c += T) dst[c] = (c == p - 1) ? sign : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * value;
        } else { // basic in root row p - w: substitute that row   (:36-42)
            const double *src = d.root_mat + (size_t)(p - w) * pitch;
            for (int c = tid; c < pcols; c += T) dst[c] = c < n ? -sign * src[c] : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * (value - d.root_rhs[p - w]);
        }
    }
    for (int i = tid; i < w + h; i += T) { // :46-52
        pos[i] = i < w + h0 ? d.root_pos[i] : i;
        var[i] = i < w + h0 ? d.root_var[i] : i;
    }
    __syncthreads();

    const WgResult out = wg_simplex<T, false>(mat, rhs, pos, var, colbuf, prow, sk, si, w, n, lp, pcols, h,
                                       wg_unit_lanes(pcols / 2, T), d.precision, d.max_pivots);
    if (LDS) { // the node's final tableau, for yalps_batch_download
        __syncthreads();
        for (int r = cg0; r < h; r += CG)
            for (int c = cu0; c < pcols; c += Uc) ws_mat[(size_t)r * pitch + c] = mat[(size_t)r * lp + c];
        for (int r = tid; r < h; r += T) ws_rhs[r] = rhs[r];
        for (int i = tid; i < w + h; i += T) {
            ws_pos[i] = pos[i];
            ws_var[i] = var[i];
        }
    }
    if (tid == 0) {
        d.status[node] = out.status;
        d.result[node] = out.result;
        d.pivots[node] = out.pivots;
        d.height[node] = h;
    }
}
