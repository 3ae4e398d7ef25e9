// yalps_hip.hip -- MI355X (gfx950 / CDNA4) dense-tableau simplex core.
//
// Replaces the body of the reference's `simplex` export (src/simplex.ts:106-144)
// behind the C ABI of include/yalps_hip.h.  Written for gfx950 only.
//
// Device-side structure (DESIGN.md has the full picture):
//   * tableau resident in HBM, row-major, rows padded to a 128-byte pitch;
//   * per pivot TWO kernels on one stream, captured 64 pairs at a time in a hipGraph:
//       select_kernel  (1 workgroup, 1024 lanes): loop control, phase-1 / phase-2 scans as
//                      64-lane wavefront arg-reductions with lowest-index tie-break, cycle
//                      detector, pivot-row normalise, and the look-ahead pricing of the NEXT
//                      entering column;
//       sweep_kernel   (chip-wide): the rank-1 fp64 row elimination (src/simplex.ts:27-38),
//                      16 B per lane coalesced, pivot-row slice held in registers, which also
//                      mirrors the next entering column and the RHS column into contiguous
//                      side arrays so the next ratio test never does a strided read;
//   * no host round trip per pivot: termination is decided on the device, later launches of a
//     batch turn into no-ops, the host polls the state once per batch.
//
// Bit-exactness contract (tests/ compare against the oracle bit for bit): separately rounded
// multiply and subtract (-ffp-contract=off, checked in the ISA: v_mul_f64 + v_add_f64, no
// v_fma_f64 in the update), IEEE division, the 1e-16 flush / skip rules of pivot(), strict
// first-wins comparisons in every scan.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yalps_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int RUNNING = -1;
constexpr int SELECT_THREADS = 1024;
constexpr int SWEEP_THREADS = 256;
constexpr int SWEEP_ROWS = 8;  // rows in flight per lane (16 B each)
constexpr int PAIRS_PER_GRAPH = 64;
// A quiet NaN with a payload no arithmetic produces: marks pivot-row entries that pivot()
// flushed to zero (src/simplex.ts:18-23), i.e. columns NOT in `nonZeroColumns`.
constexpr unsigned long long FLUSHED = 0x7FF8C0DEC0DE5EEDull;

struct alignas(16) YState {
    int32_t status;  // RUNNING or a YALPS_* status code
    int32_t phase;   // 1 | 2
    int32_t pending; // a pivot (row, col) is prepared and the sweep has to apply it
    int32_t row, col;
    int32_t la;       // look-ahead: entering column of the NEXT phase-2 iteration (0 = none)
    int32_t la_valid; // la / cbuf[cur ^ 1] were produced for the tableau as it is now
    int32_t cur;      // cbuf[cur] holds column `col` as it was before the pending pivot
    int32_t rhs_valid;
    int32_t pause; // cycle history full: host must grow it
    int32_t height;
    int32_t check_cycles;
    int64_t hist_len, hist_cap;
    int32_t *hist_leaving, *hist_entering;
    double quotient;
    double iter; // pivots done in the current phase (src/simplex.ts:69,109)
    double result;
    double precision, max_pivots;
    int64_t pivots; // total over both phases
};

struct Desc {
    double *mat;     // [hcap][pitch]
    double *cbuf[2]; // [hcap] column staging (current / next entering column)
    double *rhs;     // [hcap] mirror of column 0
    double *prow;    // [pitch] normalised pivot row, FLUSHED where pivot() wrote 0.0
    int32_t *pos, *var;
    YState *st;
    int32_t w, pitch, hcap;
};

// ------------------------------------------------------------------------------------------
// 64-lane arg-min with lowest-index tie-break (all four scans of the reference reduce to it)
// ------------------------------------------------------------------------------------------
struct KI {
    double k;
    int i;
};

__device__ __forceinline__ bool ki_better(double ka, int ia, double kb, int ib) {
    return ka < kb || (ka == kb && ia < ib);
}

__device__ __forceinline__ KI wave_argmin(KI v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_down(v.k, off, 64);
        const int oi = __shfl_down(v.i, off, 64);
        if (ki_better(ok, oi, v.k, v.i)) {
            v.k = ok;
            v.i = oi;
        }
    }
    return v;
}

// Result broadcast to every lane of the workgroup.  sk / si: 16-entry LDS scratch.
__device__ __forceinline__ KI block_argmin(KI v, double *sk, int *si) {
    v = wave_argmin(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sk[wv] = v.k;
        si[wv] = v.i;
    }
    __syncthreads();
    KI r;
    r.k = lane < nw ? sk[lane] : INFINITY;
    r.i = lane < nw ? si[lane] : INT_MAX;
    r = wave_argmin(r);
    r.k = __shfl(r.k, 0, 64);
    r.i = __shfl(r.i, 0, 64);
    return r;
}

// JS Math.round (halves toward +inf) and roundToPrecision (src/util.ts:1-4)
__host__ __device__ inline double js_round(double x) {
    if (!(fabs(x) < INFINITY)) return x; // NaN, +-inf
    const double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}
__host__ __device__ inline double round_to_precision(double num, double precision) {
    const double rounding = js_round(1.0 / precision);
    return js_round((num + 2.220446049250313e-16) * rounding) / rounding;
}

// Strided read of one tableau column into a contiguous array (only on the slow paths: first
// iteration, phase 1, or when no look-ahead column was available).
__device__ __forceinline__ void gather_column(const Desc &d, int h, int col, double *dst) {
    for (int r = threadIdx.x; r < h; r += blockDim.x) dst[r] = d.mat[(size_t)r * d.pitch + col];
    __syncthreads();
}

__device__ __forceinline__ void finish(YState *st, int status, double result) {
    if (threadIdx.x == 0) {
        st->status = status;
        st->result = result;
        st->pending = 0;
    }
}

// src/simplex.ts:44-63 -- every lane tests a set of candidate cycle lengths.
__device__ __forceinline__ bool has_cycle(YState *st, int leaving, int entering, int *flag) {
    int32_t *hl = st->hist_leaving, *he = st->hist_entering;
    const int64_t len = st->hist_len + 1;
    if (threadIdx.x == 0) {
        hl[len - 1] = leaving;
        he[len - 1] = entering;
        *flag = 0;
    }
    __syncthreads();
    bool found = false;
    for (int64_t length = 6 + threadIdx.x; length <= len / 2 && !found; length += blockDim.x) {
        bool cycle = true;
        for (int64_t i = 0; i < length; i++) {
            const int64_t item = len - 1 - i;
            if (hl[item] != hl[item - length] || he[item] != he[item - length]) {
                cycle = false;
                break;
            }
        }
        found = cycle;
    }
    if (found) *flag = 1;
    __syncthreads();
    return *flag != 0;
}

// Pivot bookkeeping + pivot-row normalise (src/simplex.ts:6-25) for the pivot (row, col);
// cb = column `col` before the pivot.  With lookahead (phase 2) it also prices the objective
// row as it will be AFTER this pivot and returns the next entering column (0 = none).
__device__ __forceinline__ int prepare_pivot(const Desc &d, YState *st, int row, int col, const double *cb,
                                             bool lookahead, double precision, double *sk, int *si) {
    const int w = d.w;
    const double q = cb[row];
    double *mrow = d.mat + (size_t)row * d.pitch;
    const double *m0 = d.mat;
    const double coef0 = cb[0];
    const bool touched0 = fabs(coef0) > 1e-16;
    const double inv_q = 1.0 / q;
    const double neg0 = -coef0 / q;
    KI best = {INFINITY, INT_MAX};
    for (int c = threadIdx.x; c < w; c += blockDim.x) {
        const double v = mrow[c];
        const bool nz = fabs(v) > 1e-16;
        const double pn = nz ? v / q : 0.0;
        mrow[c] = (c == col) ? inv_q : pn;
        d.prow[c] = nz ? pn : __longlong_as_double((long long)FLUSHED);
        if (lookahead && c >= 1) {
            double o = m0[c];
            if (touched0) {
                if (c == col)
                    o = neg0;
                else if (nz) {
                    const double prod = coef0 * pn;
                    o = o - prod;
                }
            }
            if (o > precision && ki_better(-o, c, best.k, best.i)) {
                best.k = -o;
                best.i = c;
            }
        }
    }
    if (threadIdx.x == 0) {
        // basis bookkeeping, src/simplex.ts:7-12
        const int leaving = d.var[w + row], entering = d.var[col];
        d.var[w + row] = entering;
        d.var[col] = leaving;
        d.pos[leaving] = col;
        d.pos[entering] = w + row;
    }
    int la = 0;
    if (lookahead) {
        const KI r = block_argmin(best, sk, si);
        la = r.i == INT_MAX ? 0 : r.i;
    } else {
        __syncthreads();
    }
    // the pivot row is skipped by the sweep: mirror its new entries here
    if (threadIdx.x == 0) {
        d.rhs[row] = mrow[0];
        if (la > 0) d.cbuf[st->cur ^ 1][row] = mrow[la];
        st->quotient = q;
    }
    return la;
}

// ------------------------------------------------------------------------------------------
// select_kernel: one workgroup; everything of phase1()/phase2() except the elimination
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SELECT_THREADS) void select_kernel(Desc d) {
    __shared__ double sk[16];
    __shared__ int si[16];
    __shared__ int cyc_flag;
    YState *st = d.st;
    if (st->status != RUNNING || st->pause) return;
    const int h = st->height, w = d.w;
    const double precision = st->precision, max_pivots = st->max_pivots;
    int phase = st->phase, cur = st->cur;
    double iter = st->iter;
    bool la_valid = st->la_valid != 0;
    const int la = st->la;
    const int tid = threadIdx.x, nt = blockDim.x;

    if (!st->rhs_valid) gather_column(d, h, 0, d.rhs);

    int row = 0, col = 0;
    for (;;) {
        if (!(iter < max_pivots)) { // src/simplex.ts:69,109 loop bound; :102,141
            finish(st, YALPS_CYCLED, NAN);
            return;
        }
        if (phase == 1) {
            // leaving row: most negative RHS, strict <, first wins (src/simplex.ts:111-119)
            KI b = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += nt) {
                const double v = d.rhs[r];
                if (v < -precision && ki_better(v, r, b.k, b.i)) {
                    b.k = v;
                    b.i = r;
                }
            }
            b = block_argmin(b, sk, si);
            if (b.i == INT_MAX) { // :120 tail call of phase2 -- fresh counter and history
                phase = 2;
                iter = 0.0;
                la_valid = false;
                if (tid == 0) st->hist_len = 0;
                continue;
            }
            row = b.i;
            // entering column: max -M[0,c]/M[row,c] over M[row,c] < -precision (:123-134)
            const double *mrow = d.mat + (size_t)row * d.pitch;
            KI e = {INFINITY, INT_MAX};
            for (int c = 1 + tid; c < w; c += nt) {
                const double coefficient = mrow[c];
                if (coefficient < -precision) {
                    const double ratio = -d.mat[c] / coefficient;
                    if (ratio > -INFINITY && ki_better(-ratio, c, e.k, e.i)) {
                        e.k = -ratio;
                        e.i = c;
                    }
                }
            }
            e = block_argmin(e, sk, si);
            if (e.i == INT_MAX) { // :135
                finish(st, YALPS_INFEASIBLE, NAN);
                return;
            }
            col = e.i;
            gather_column(d, h, col, d.cbuf[cur]);
            break;
        } else {
            // entering column: Dantzig pricing (src/simplex.ts:71-79), normally already known
            if (la_valid) {
                col = la;
                cur ^= 1;
            } else {
                KI p = {INFINITY, INT_MAX};
                for (int c = 1 + tid; c < w; c += nt) {
                    const double rc = d.mat[c];
                    if (rc > precision && ki_better(-rc, c, p.k, p.i)) {
                        p.k = -rc;
                        p.i = c;
                    }
                }
                p = block_argmin(p, sk, si);
                col = p.i == INT_MAX ? 0 : p.i;
                if (col) gather_column(d, h, col, d.cbuf[cur]);
            }
            if (col == 0) { // :80
                finish(st, YALPS_OPTIMAL, round_to_precision(d.mat[0], precision));
                return;
            }
            // leaving row: min-ratio test with the early break (:83-95).  Closed form: the
            // lowest-index eligible row whose ratio is <= precision if there is one (key -inf),
            // else the lowest-index arg-min.
            const double *cb = d.cbuf[cur];
            KI b = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += nt) {
                const double value = cb[r];
                if (value <= precision) continue;
                const double ratio = d.rhs[r] / value;
                if (!(ratio < INFINITY)) continue;
                const double key = (ratio <= precision) ? -INFINITY : ratio;
                if (ki_better(key, r, b.k, b.i)) {
                    b.k = key;
                    b.i = r;
                }
            }
            b = block_argmin(b, sk, si);
            if (b.i == INT_MAX) { // :96
                finish(st, YALPS_UNBOUNDED, (double)col);
                return;
            }
            row = b.i;
            break;
        }
    }

    if (st->check_cycles) { // :98,137
        if (st->hist_len >= st->hist_cap) {
            // history full: leave the state untouched except for the phase switch, which is
            // idempotent, and let the host grow the buffers
            if (tid == 0) {
                st->pause = 1;
                st->pending = 0;
                if (phase != st->phase) {
                    st->phase = phase;
                    st->iter = iter;
                    st->la_valid = 0;
                }
            }
            return;
        }
        const bool cyc = has_cycle(st, d.var[w + row], d.var[col], &cyc_flag);
        if (tid == 0) st->hist_len = st->hist_len + 1;
        if (cyc) {
            finish(st, YALPS_CYCLED, NAN);
            return;
        }
    }

    if (tid == 0) st->cur = cur; // prepare_pivot mirrors into cbuf[cur ^ 1]
    __syncthreads();
    const int next = prepare_pivot(d, st, row, col, d.cbuf[cur], phase == 2, precision, sk, si);
    if (tid == 0) {
        st->phase = phase;
        st->row = row;
        st->col = col;
        st->la = next;
        st->la_valid = phase == 2;
        st->rhs_valid = 1;
        st->iter = iter + 1.0;
        st->pivots = st->pivots + 1;
        st->pending = 1;
    }
}

// One explicit pivot (yalps_tableau_pivot / bench): same preparation, no scans.
__global__ __launch_bounds__(SELECT_THREADS) void prepare_kernel(Desc d, int row, int col) {
    __shared__ double sk[16];
    __shared__ int si[16];
    YState *st = d.st;
    gather_column(d, st->height, col, d.cbuf[st->cur]);
    prepare_pivot(d, st, row, col, d.cbuf[st->cur], false, 0.0, sk, si);
    if (threadIdx.x == 0) {
        st->row = row;
        st->col = col;
        st->la = 0;
        st->la_valid = 0;
        st->pending = 1;
    }
}

// ------------------------------------------------------------------------------------------
// sweep_kernel: row elimination, src/simplex.ts:27-38.  HBM-bound rank-1 update.
// grid = (column blocks of 512 doubles, row groups); lane = one 16-byte unit of a row; a block
// walks rows rg, rg + RG, ... with SWEEP_ROWS loads in flight per lane.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SWEEP_THREADS) void sweep_kernel(Desc d, int force) {
    const YState *st = d.st;
    if (!force && (st->status != RUNNING || !st->pending)) return;
    const int c0 = (blockIdx.x * SWEEP_THREADS + threadIdx.x) * 2;
    if (c0 >= d.w) return;
    const int h = st->height, row = st->row, col = st->col;
    const int la = (st->la_valid && st->la > 0) ? st->la : -1;
    const double q = st->quotient;
    const int cur = st->cur;
    const double *__restrict__ ccol = d.cbuf[cur];
    double *__restrict__ ncol = d.cbuf[cur ^ 1];
    const int pitch = d.pitch;

    const double2 p = *reinterpret_cast<const double2 *>(d.prow + c0);
    const bool f0 = (unsigned long long)__double_as_longlong(p.x) != FLUSHED;
    const bool f1 = (unsigned long long)__double_as_longlong(p.y) != FLUSHED;
    const bool has_col = (col >> 1) == (c0 >> 1);
    const bool block_has_col = (col >> 9) == (int)blockIdx.x;
    const bool has_la = la >= 0 && (la >> 1) == (c0 >> 1);
    const int RG = gridDim.y;

    for (int rbase = blockIdx.y; rbase < h; rbase += RG * SWEEP_ROWS) {
        double coef[SWEEP_ROWS];
        bool live[SWEEP_ROWS], act[SWEEP_ROWS];
        double2 v[SWEEP_ROWS];
#pragma unroll
        for (int g = 0; g < SWEEP_ROWS; g++) {
            const int r = rbase + g * RG;
            live[g] = r < h && r != row;
            coef[g] = live[g] ? ccol[r] : 0.0;
            act[g] = live[g] && fabs(coef[g]) > 1e-16; // rows with |coef| <= 1e-16 stay untouched
        }
#pragma unroll
        for (int g = 0; g < SWEEP_ROWS; g++) {
            const int r = rbase + g * RG;
            if (act[g]) v[g] = *reinterpret_cast<const double2 *>(d.mat + (size_t)r * pitch + c0);
        }
#pragma unroll
        for (int g = 0; g < SWEEP_ROWS; g++) {
            const int r = rbase + g * RG;
            if (act[g]) {
                double *mp = d.mat + (size_t)r * pitch + c0;
                const double c = coef[g];
                double2 o = v[g];
                if (f0) {
                    const double prod = c * p.x;
                    o.x = o.x - prod;
                }
                if (f1) {
                    const double prod = c * p.y;
                    o.y = o.y - prod;
                }
                if (block_has_col) {
                    const double nq = -c / q;
                    if (has_col) {
                        if (col & 1)
                            o.y = nq;
                        else
                            o.x = nq;
                    }
                }
                *reinterpret_cast<double2 *>(mp) = o;
                if (c0 == 0) d.rhs[r] = o.x;
                if (has_la) ncol[r] = (la & 1) ? o.y : o.x;
            } else if (live[g] && has_la) {
                ncol[r] = d.mat[(size_t)r * pitch + la];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? YALPS_E_NOMEM : YALPS_E_DEVICE,               \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

} // namespace

struct yalps_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool eager = false;
};

struct yalps_tableau {
    yalps_ctx *ctx = nullptr;
    Desc d{};
    int32_t height = 0;
    dim3 sweep_grid;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    YState *host_state = nullptr; // pinned, 4 rotating slots
    hipEvent_t slot_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *hist[2] = {nullptr, nullptr};
    int64_t hist_cap = 0;
};

namespace {

int launch_pair_batch(yalps_tableau *t, int pairs) {
    hipStream_t s = t->ctx->stream;
    for (int i = 0; i < pairs; i++) {
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SELECT_THREADS), 0, s, t->d);
        hipLaunchKernelGGL(sweep_kernel, t->sweep_grid, dim3(SWEEP_THREADS), 0, s, t->d, 0);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ensure_graph(yalps_tableau *t) {
    if (t->graph_exec || t->ctx->eager) return 0;
    hipStream_t s = t->ctx->stream;
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = launch_pair_batch(t, PAIRS_PER_GRAPH);
    hipError_t e = hipStreamEndCapture(s, &t->graph);
    if (rc) return rc;
    HIP_TRY(e);
    HIP_TRY(hipGraphInstantiate(&t->graph_exec, t->graph, nullptr, nullptr, 0));
    return 0;
}

int run_batch(yalps_tableau *t) {
    if (t->ctx->eager) return launch_pair_batch(t, PAIRS_PER_GRAPH);
    HIP_TRY(hipGraphLaunch(t->graph_exec, t->ctx->stream));
    return 0;
}

int grow_history(yalps_tableau *t, int64_t need, const YState *cur_state) {
    int64_t cap = t->hist_cap ? t->hist_cap : 4096;
    while (cap < need) cap *= 2;
    if (cap == t->hist_cap) return 0;
    hipStream_t s = t->ctx->stream;
    for (int k = 0; k < 2; k++) {
        int32_t *nb = nullptr;
        HIP_TRY(hipMalloc(&nb, sizeof(int32_t) * (size_t)cap));
        if (t->hist[k] && cur_state && cur_state->hist_len > 0)
            HIP_TRY(hipMemcpyAsync(nb, t->hist[k], sizeof(int32_t) * (size_t)cur_state->hist_len,
                                   hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (t->hist[k]) HIP_TRY(hipFree(t->hist[k]));
        t->hist[k] = nb;
    }
    t->hist_cap = cap;
    return 0;
}

} // namespace

extern "C" {

const char *yalps_last_error(void) { return g_err.c_str(); }

int32_t yalps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

double yalps_round_to_precision(double num, double precision) { return round_to_precision(num, precision); }

void yalps_dense_lp_f64(int32_t M, int32_t N, double seed, double *matrix) {
    // tests/helpers/util.ts:20-41 of the reference: seed is a double, += 0x9e3779b9 unwrapped
    auto next = [&seed]() {
        seed += 2654435769.0;
        uint32_t x = (uint32_t)(uint64_t)fmod(seed, 4294967296.0);
        x ^= x >> 16;
        x *= 0x21f0aaadu;
        x ^= x >> 15;
        x *= 0xd35a2d97u;
        x ^= x >> 15;
        return (double)x / 4294967296.0;
    };
    const int32_t w = N + 1, h = M + 1;
    matrix[0] = 0.0;
    for (int32_t j = 1; j < w; j++) matrix[j] = next();
    for (int32_t r = 1; r < h; r++) {
        double *mr = matrix + (size_t)r * w;
        mr[0] = (double)N * 0.25 * (1.0 + next());
        for (int32_t j = 1; j < w; j++) mr[j] = next();
    }
}

int32_t yalps_ctx_create(int32_t device, yalps_ctx **out) {
    if (!out) return fail(YALPS_E_ARG, "yalps_ctx_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(YALPS_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(YALPS_E_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(YALPS_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
    yalps_ctx *c = new yalps_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    const char *e = std::getenv("YALPS_HIP_EAGER");
    c->eager = e && *e && *e != '0';
    *out = c;
    return 0;
}

void yalps_ctx_destroy(yalps_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t yalps_tableau_create(yalps_ctx *ctx, int32_t width, int32_t hcap, yalps_tableau **out) {
    if (!ctx || !out || width < 1 || hcap < 1) return fail(YALPS_E_ARG, "yalps_tableau_create: bad argument");
    if ((int64_t)width + hcap > INT32_MAX / 2) return fail(YALPS_E_ARG, "tableau too large");
    HIP_TRY(hipSetDevice(ctx->device));
    yalps_tableau *t = new yalps_tableau();
    t->ctx = ctx;
    Desc &d = t->d;
    d.w = width;
    d.hcap = hcap;
    d.pitch = (width + 15) / 16 * 16; // 128-byte rows
    const size_t mat_bytes = sizeof(double) * (size_t)d.pitch * hcap;
    HIP_TRY(hipMalloc(&d.mat, mat_bytes));
    HIP_TRY(hipMemsetAsync(d.mat, 0, mat_bytes, ctx->stream));
    for (int k = 0; k < 2; k++) HIP_TRY(hipMalloc(&d.cbuf[k], sizeof(double) * (size_t)hcap));
    HIP_TRY(hipMalloc(&d.rhs, sizeof(double) * (size_t)hcap));
    HIP_TRY(hipMalloc(&d.prow, sizeof(double) * (size_t)d.pitch));
    HIP_TRY(hipMemsetAsync(d.prow, 0, sizeof(double) * (size_t)d.pitch, ctx->stream));
    HIP_TRY(hipMalloc(&d.pos, sizeof(int32_t) * (size_t)(width + hcap)));
    HIP_TRY(hipMalloc(&d.var, sizeof(int32_t) * (size_t)(width + hcap)));
    HIP_TRY(hipMalloc(&d.st, sizeof(YState)));
    HIP_TRY(hipMemsetAsync(d.st, 0, sizeof(YState), ctx->stream));
    HIP_TRY(hipHostMalloc(&t->host_state, sizeof(YState) * 4, hipHostMallocDefault));
    for (int k = 0; k < 4; k++) HIP_TRY(hipEventCreateWithFlags(&t->slot_ev[k], hipEventDisableTiming));
    // sweep grid: column blocks x row groups, ~8 rows per lane, capped near 2048 workgroups
    const int col_blocks = ((width + 1) / 2 + SWEEP_THREADS - 1) / SWEEP_THREADS;
    int rg = (hcap + SWEEP_ROWS - 1) / SWEEP_ROWS;
    const int max_rg = (2048 + col_blocks - 1) / col_blocks;
    if (rg > max_rg) rg = max_rg;
    if (rg < 1) rg = 1;
    t->sweep_grid = dim3(col_blocks, rg, 1);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = t;
    return 0;
}

void yalps_tableau_destroy(yalps_tableau *t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->graph_exec) (void)hipGraphExecDestroy(t->graph_exec);
    if (t->graph) (void)hipGraphDestroy(t->graph);
    Desc &d = t->d;
    void *bufs[] = {d.mat, d.cbuf[0], d.cbuf[1], d.rhs, d.prow, d.pos, d.var, d.st, t->hist[0], t->hist[1]};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (t->host_state) (void)hipHostFree(t->host_state);
    for (auto &e : t->slot_ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

int32_t yalps_tableau_height(const yalps_tableau *t) { return t ? t->height : 0; }

int32_t yalps_tableau_upload(yalps_tableau *t, const double *matrix, int32_t height, const int32_t *pos,
                             const int32_t *var) {
    if (!t || !matrix || !pos || !var) return fail(YALPS_E_ARG, "yalps_tableau_upload: NULL argument");
    if (height < 1 || height > t->d.hcap) return fail(YALPS_E_ARG, "yalps_tableau_upload: height exceeds capacity");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    const Desc &d = t->d;
    HIP_TRY(hipMemcpy2DAsync(d.mat, sizeof(double) * d.pitch, matrix, sizeof(double) * d.w, sizeof(double) * d.w,
                             height, hipMemcpyHostToDevice, s));
    const size_t nperm = sizeof(int32_t) * (size_t)(d.w + height);
    HIP_TRY(hipMemcpyAsync(d.pos, pos, nperm, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d.var, var, nperm, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    t->height = height;
    return 0;
}

int32_t yalps_tableau_download(yalps_tableau *t, double *matrix, int32_t *pos, int32_t *var) {
    if (!t) return fail(YALPS_E_ARG, "yalps_tableau_download: NULL tableau");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    const Desc &d = t->d;
    if (matrix)
        HIP_TRY(hipMemcpy2DAsync(matrix, sizeof(double) * d.w, d.mat, sizeof(double) * d.pitch,
                                 sizeof(double) * d.w, t->height, hipMemcpyDeviceToHost, s));
    const size_t nperm = sizeof(int32_t) * (size_t)(d.w + t->height);
    if (pos) HIP_TRY(hipMemcpyAsync(pos, d.pos, nperm, hipMemcpyDeviceToHost, s));
    if (var) HIP_TRY(hipMemcpyAsync(var, d.var, nperm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_tableau_download_rhs(yalps_tableau *t, double *col0) {
    if (!t || !col0) return fail(YALPS_E_ARG, "yalps_tableau_download_rhs: NULL argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    const Desc &d = t->d;
    HIP_TRY(hipMemcpy2DAsync(col0, sizeof(double), d.mat, sizeof(double) * d.pitch, sizeof(double), t->height,
                             hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_tableau_copy(yalps_tableau *dst, const yalps_tableau *src) {
    if (!dst || !src) return fail(YALPS_E_ARG, "yalps_tableau_copy: NULL argument");
    if (dst->d.w != src->d.w || dst->d.hcap < src->height || dst->ctx != src->ctx)
        return fail(YALPS_E_ARG, "yalps_tableau_copy: incompatible tableaux");
    HIP_TRY(hipSetDevice(dst->ctx->device));
    hipStream_t s = dst->ctx->stream;
    HIP_TRY(hipMemcpyAsync(dst->d.mat, src->d.mat, sizeof(double) * (size_t)src->d.pitch * src->height,
                           hipMemcpyDeviceToDevice, s));
    const size_t nperm = sizeof(int32_t) * (size_t)(src->d.w + src->height);
    HIP_TRY(hipMemcpyAsync(dst->d.pos, src->d.pos, nperm, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst->d.var, src->d.var, nperm, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    dst->height = src->height;
    return 0;
}

static int32_t init_state(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles) {
    hipStream_t s = t->ctx->stream;
    if (checkCycles && !t->hist_cap) {
        int rc = grow_history(t, 4096, nullptr);
        if (rc) return rc;
    }
    YState *hs = &t->host_state[0];
    std::memset(hs, 0, sizeof(YState));
    hs->status = RUNNING;
    hs->phase = 1;
    hs->height = t->height;
    hs->check_cycles = checkCycles ? 1 : 0;
    hs->hist_cap = t->hist_cap;
    hs->hist_leaving = t->hist[0];
    hs->hist_entering = t->hist[1];
    hs->precision = precision;
    hs->max_pivots = maxPivots;
    hs->result = NAN;
    HIP_TRY(hipMemcpyAsync(t->d.st, hs, sizeof(YState), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // slot 0 is reused below
    return 0;
}

int32_t yalps_tableau_solve(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles,
                            double *result_out, int64_t *pivots_out, float *gpu_ms_out) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_tableau_solve: no tableau uploaded");
    yalps_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = init_state(t, precision, maxPivots, checkCycles);
    if (rc) return rc;
    rc = ensure_graph(t);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev0, s));
    // keep one batch in flight while the previous batch's state is inspected
    YState fin;
    int issued = 0, checked = 0;
    for (;;) {
        rc = run_batch(t);
        if (rc) return rc;
        const int slot = issued & 3;
        HIP_TRY(hipMemcpyAsync(&t->host_state[slot], t->d.st, sizeof(YState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(t->slot_ev[slot], s));
        issued++;
        if (issued - checked < 2) continue;
        const int cs = checked & 3;
        HIP_TRY(hipEventSynchronize(t->slot_ev[cs]));
        checked++;
        const YState &hs = t->host_state[cs];
        if (hs.status != RUNNING) {
            fin = hs;
            break;
        }
        if (hs.pause) {
            // drain, grow the cycle history, resume
            HIP_TRY(hipStreamSynchronize(s));
            YState now;
            HIP_TRY(hipMemcpy(&now, t->d.st, sizeof(YState), hipMemcpyDeviceToHost));
            checked = issued;
            if (now.status != RUNNING) {
                fin = now;
                break;
            }
            rc = grow_history(t, now.hist_cap * 2, &now);
            if (rc) return rc;
            now.pause = 0;
            now.hist_cap = t->hist_cap;
            now.hist_leaving = t->hist[0];
            now.hist_entering = t->hist[1];
            HIP_TRY(hipMemcpy(t->d.st, &now, sizeof(YState), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (gpu_ms_out) HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    if (result_out) *result_out = fin.result;
    if (pivots_out) *pivots_out = fin.pivots;
    return fin.status;
}

int32_t yalps_tableau_pivot(yalps_tableau *t, int32_t row, int32_t col) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_tableau_pivot: no tableau uploaded");
    if (row < 0 || row >= t->height || col < 0 || col >= t->d.w) return fail(YALPS_E_ARG, "pivot out of range");
    HIP_TRY(hipSetDevice(t->ctx->device));
    int rc = init_state(t, 1e-8, 0.0, 0);
    if (rc) return rc;
    hipStream_t s = t->ctx->stream;
    hipLaunchKernelGGL(prepare_kernel, dim3(1), dim3(SELECT_THREADS), 0, s, t->d, row, col);
    hipLaunchKernelGGL(sweep_kernel, t->sweep_grid, dim3(SWEEP_THREADS), 0, s, t->d, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_tableau_bench_sweep(yalps_tableau *t, int32_t row, int32_t col, int32_t launches, float *avg_us_out) {
    if (!t || t->height < 1 || launches < 1) return fail(YALPS_E_ARG, "yalps_tableau_bench_sweep: bad argument");
    if (row < 0 || row >= t->height || col < 0 || col >= t->d.w) return fail(YALPS_E_ARG, "pivot out of range");
    yalps_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = init_state(t, 1e-8, 0.0, 0);
    if (rc) return rc;
    hipStream_t s = c->stream;
    hipLaunchKernelGGL(prepare_kernel, dim3(1), dim3(SELECT_THREADS), 0, s, t->d, row, col);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(sweep_kernel, t->sweep_grid, dim3(SWEEP_THREADS), 0, s, t->d, 1);
    HIP_TRY(hipEventRecord(c->ev0, s));
    for (int i = 0; i < launches; i++)
        hipLaunchKernelGGL(sweep_kernel, t->sweep_grid, dim3(SWEEP_THREADS), 0, s, t->d, 1);
    HIP_TRY(hipEventRecord(c->ev1, s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (avg_us_out) *avg_us_out = ms * 1000.f / (float)launches;
    return 0;
}

// ---- the drop-in entry point -------------------------------------------------------------
namespace {
std::mutex g_default_mu;
yalps_ctx *g_default_ctx = nullptr;
yalps_tableau *g_default_tab = nullptr;
} // namespace

int32_t yalps_simplex_f64_ex(double *matrix, int32_t width, int32_t height, int32_t *pos, int32_t *var,
                             double precision, double maxPivots, int32_t checkCycles, int32_t copyback,
                             double *result_out, int64_t *pivots_out) {
    if (!matrix || !pos || !var || width < 1 || height < 1)
        return fail(YALPS_E_ARG, "yalps_simplex_f64: bad argument");
    std::lock_guard<std::mutex> lock(g_default_mu);
    int rc;
    if (!g_default_ctx) {
        const char *dev = std::getenv("YALPS_HIP_DEVICE");
        rc = yalps_ctx_create(dev ? std::atoi(dev) : 0, &g_default_ctx);
        if (rc) return rc;
    }
    yalps_tableau *t = g_default_tab;
    if (!t || t->d.w != width || t->d.hcap < height) {
        if (t) yalps_tableau_destroy(t);
        g_default_tab = nullptr;
        rc = yalps_tableau_create(g_default_ctx, width, height, &t);
        if (rc) return rc;
        g_default_tab = t;
    }
    rc = yalps_tableau_upload(t, matrix, height, pos, var);
    if (rc) return rc;
    const int32_t status = yalps_tableau_solve(t, precision, maxPivots, checkCycles, result_out, pivots_out, nullptr);
    if (status < 0) return status;
    if (copyback == YALPS_COPYBACK_SOLUTION) {
        std::vector<double> col0((size_t)height);
        rc = yalps_tableau_download_rhs(t, col0.data());
        if (rc) return rc;
        for (int32_t r = 0; r < height; r++) matrix[(size_t)r * width] = col0[(size_t)r];
        rc = yalps_tableau_download(t, nullptr, pos, var);
    } else {
        rc = yalps_tableau_download(t, matrix, pos, var);
    }
    if (rc) return rc;
    return status;
}

int32_t yalps_simplex_f64(double *matrix, int32_t width, int32_t height, int32_t *pos, int32_t *var,
                          double precision, double maxPivots, int32_t checkCycles, double *result_out) {
    return yalps_simplex_f64_ex(matrix, width, height, pos, var, precision, maxPivots, checkCycles,
                                YALPS_COPYBACK_FULL, result_out, nullptr);
}

} // extern "C"
