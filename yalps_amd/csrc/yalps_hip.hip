// yalps_hip.hip -- MI355X (gfx950 / CDNA4) dense-tableau simplex core.
//
// Replaces the body of the reference's `simplex` export (src/simplex.ts:106-144)
// behind the C ABI of include/yalps_hip.h.  Written for gfx950 only.
//
// Device-side structure (DESIGN.md has the full picture):
//   * tableau resident in HBM.  Column 0 (the RHS column) lives in its own contiguous array
//     `rhs[h]`; the variable columns 1..w-1 live row-major in `mat[h][pitch]` (128-byte rows),
//     so every row is a whole number of 16-byte lane units and the ratio-test inputs are
//     contiguous.
//   * ONE kernel launch per pivot (pivot_kernel, mode FUSED), 64 launches per hipGraph replay:
//       - every workgroup spans the full row width (lane = 16-byte unit(s) of a row) and owns
//         the rows b, b+NB, b+2NB, ...;
//       - prologue, redundantly in every workgroup: reduce the per-workgroup partials the
//         previous launch left (min-ratio candidates / most-negative-RHS candidates) with
//         64-lane arg-min reductions (lowest index wins ties) -> leaving row; fetch that row,
//         normalise it in registers (src/simplex.ts:14-25) and price the objective row as it
//         will be after this pivot -> next entering column (look-ahead);
//       - body: rank-1 fp64 elimination of the workgroup's rows (src/simplex.ts:27-38), all row
//         loads of a batch in flight at once; the updated entries of the next entering column
//         and of the RHS are picked out of the registers they already sit in and reduced to
//         this workgroup's partial for the next launch.
//     The only grid-wide dependency (arg-min over all rows) is carried by the kernel boundary.
//   * checkCycles=true runs the same kernel as alternating DECIDE (one workgroup: selection +
//     the reference's cycle detector) and APPLY launches.
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
constexpr int MODE_FUSED = 0, MODE_DECIDE = 1, MODE_APPLY = 2, MODE_SHARD = 3;
constexpr int SHARD_HDR = 8;  // doubles in front of the two candidate rows of a gather slot
constexpr int MAX_SHARDS = 8; // one node of MI355X
constexpr int LAUNCHES_PER_GRAPH = 64; // even: state parity returns to 0 after a replay
constexpr int MAX_BLOCKS = 1024;       // partial arrays / reduction width
// A quiet NaN with a payload no arithmetic produces: marks pivot-row entries that pivot() flushed to
// zero (src/simplex.ts:18-23, i.e. columns NOT in `nonZeroColumns`) where the row is staged in LDS.
constexpr unsigned long long FLUSHED = 0x7FF8C0DEC0DE5EEDull;

// Per-solve constants (host-written once per solve; the cycle-history pointers again on growth).
struct alignas(16) YConst {
    int32_t height;
    int32_t check_cycles;
    int64_t hist_cap;
    int32_t *hist_leaving, *hist_entering;
    double precision, max_pivots;
};

// Dynamic solver state, ping-ponged between launches.  The hot path writes every field from
// registers (no read-modify-write chain at the end of a launch).
struct alignas(16) YState {
    int32_t status;    // RUNNING or a YALPS_* status code
    int32_t phase;     // 1 | 2
    int32_t bootstrap; // no partials exist yet: next APPLY/FUSED launch only scans
    int32_t la;        // column whose min-ratio partials are in part_ratio[pbuf] (0 = none priced)
    int32_t pbuf;      // which partial buffers the next launch reads
    int32_t mbuf;      // which tableau buffer holds the current tableau (the other one is written)
    int32_t pause;     // cycle history full: host must grow it
    int32_t dec_valid; // DECIDE -> APPLY hand-off
    int32_t dec_row, dec_col;
    // basis bookkeeping (src/simplex.ts:7-12) of the pivot just applied, carried out by the NEXT
    // launch (its loads are then the oldest of that launch instead of the last of this one)
    int32_t swap_valid, swap_row, swap_col;
    int32_t pad_;
    int64_t hist_len;
    double iter; // pivots done in the current phase (src/simplex.ts:69,109)
    double result;
    int64_t pivots; // total over both phases
};

struct alignas(16) Part {
    double key;
    int32_t idx;
    int32_t pad_;
};

struct Desc {
    // The tableau is ping-ponged: a pivot reads buffer [mbuf] and writes buffer [mbuf ^ 1], so no
    // workgroup ever reads a row (pivot row, objective row, pivot column) that another workgroup
    // of the same launch is overwriting.
    double *mat[2]; // [hcap][pitch]: columns 1..w-1 of the reference tableau
    double *rhs[2]; // [hcap]: column 0
    int32_t *pos, *var;
    YState *st;          // [2], ping-pong by launch parity
    YConst *cst;
    Part *part_ratio[2]; // [MAX_BLOCKS] each
    Part *part_rhs[2];
    int32_t w, n, pitch, hcap; // n = w - 1 variable columns
    int32_t nb;                // workgroups of an APPLY/FUSED launch = row stride = number of partials
    // row sharding over GPUs: this rank holds the objective row (local row 0,
    // replicated) + global rows [bounds[rank], bounds[rank+1]) as local rows 1..; a local row
    // r >= 1 is global row r + row_base.  Unsharded: nshards = 1, row_base = 0.
    int32_t nshards, shard_rank, row_base;
    int32_t bounds[MAX_SHARDS + 1];
    // resident (on-chip) solver: per-workgroup candidate hand-off buffers, ping-pong by epoch parity
    double *rc_rows[2];             // [nb][pitch] candidate row of each workgroup
    double *rc_key[2];              // [nb] RHS entry of each workgroup's candidate row
    unsigned long long *rc_flag[2]; // [nb][2] {candidate key bits, (epoch << 32) | global row index}
    int32_t *rc_err;                // set when a workgroup gives up waiting (never expected)
    int32_t perm_len;
};

// ------------------------------------------------------------------------------------------
// 64-lane arg-min with lowest-index tie-break (all four scans of the reference reduce to it).
// Built on DPP lane permutes (VALU speed); __shfl_* would go through ds_bpermute, ~1 us per
// 64-lane (double,int) reduction, which was most of a pivot's fixed cost.
// ------------------------------------------------------------------------------------------
struct KI {
    double k;
    int i;
};

__device__ __forceinline__ bool ki_better(double ka, int ia, double kb, int ib) {
    return ka < kb || (ka == kb && ia < ib);
}

// DPP controls: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// every lane of a 16-lane row gets the row's minimum (keys are never NaN)
__device__ __forceinline__ double row16_min(double v) {
    v = fmin(v, dpp_f64<DPP_XOR1>(v));
    v = fmin(v, dpp_f64<DPP_XOR2>(v));
    v = fmin(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = fmin(v, dpp_f64<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
    v = min(v, dpp_i32<DPP_XOR1>(v));
    v = min(v, dpp_i32<DPP_XOR2>(v));
    v = min(v, dpp_i32<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_i32<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ double lane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ KI row16_argmin(KI v) {
    KI r;
    r.k = row16_min(v.k);
    r.i = row16_min(v.k == r.k ? v.i : INT_MAX);
    return r;
}
// result uniform over the wave
__device__ __forceinline__ KI wave_argmin(KI v) {
    const double m = row16_min(v.k);
    KI r;
    r.k = fmin(fmin(lane_f64(m, 0), lane_f64(m, 16)), fmin(lane_f64(m, 32), lane_f64(m, 48)));
    const int i = row16_min(v.k == r.k ? v.i : INT_MAX);
    r.i = min(min(__builtin_amdgcn_readlane(i, 0), __builtin_amdgcn_readlane(i, 16)),
              min(__builtin_amdgcn_readlane(i, 32), __builtin_amdgcn_readlane(i, 48)));
    return r;
}

// Result broadcast to every lane of the workgroup.  sk / si: [2][16] LDS scratch, `slot`
// alternates between consecutive calls (so one barrier per call is enough).
template <int T>
__device__ __forceinline__ KI block_argmin(KI v, double (*sk)[16], int (*si)[16], int slot) {
    constexpr int NW = T / 64;
    v = wave_argmin(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sk[slot][wv] = v.k;
        si[slot][wv] = v.i;
    }
    __syncthreads();
    KI r;
    r.k = (lane & 15) < NW ? sk[slot][lane & 15] : INFINITY;
    r.i = (lane & 15) < NW ? si[slot][lane & 15] : INT_MAX;
    return row16_argmin(r); // every 16-lane row holds all NW wave results
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

// src/simplex.ts:44-63 -- every lane tests a set of candidate cycle lengths (DECIDE launches).
__device__ __forceinline__ bool has_cycle(const YConst *C, int64_t hist_len, int leaving, int entering, int *flag) {
    int32_t *hl = C->hist_leaving, *he = C->hist_entering;
    const int64_t len = hist_len + 1;
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

// Sout = Sin, 16 bytes at a time, straight from global to global (a `YState s = *Sin` local copy
// is turned into a per-lane LDS array by hipcc).
__device__ __forceinline__ void state_copy(YState *dst, const YState *src) {
    static_assert(sizeof(YState) % 16 == 0, "YState is copied as int4 words");
    const int4 *s4 = reinterpret_cast<const int4 *>(src);
    int4 *d4 = reinterpret_cast<int4 *>(dst);
#pragma unroll
    for (unsigned i = 0; i < sizeof(YState) / 16; i++) d4[i] = s4[i];
}

// 16-byte row load; nt = non-temporal (streaming) cache policy
__device__ __forceinline__ double2 ld_row(const double *p, bool nt) {
    if (nt) return make_double2(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1));
    return *reinterpret_cast<const double2 *>(p);
}

// By-value selects: a reference + runtime element index would turn into a dynamically indexed
// private array, which hipcc places in scratch / LDS instead of registers.
__device__ __forceinline__ double elem(double2 v, int e) {
    const double a = v.x, b = v.y;
    return e ? b : a;
}
__device__ __forceinline__ double2 with_elem(double2 v, int e, double x) {
    return make_double2(e ? v.x : x, e ? x : v.y);
}

// ------------------------------------------------------------------------------------------
// pivot_kernel<T lanes, J units per lane per row, R rows per lane and batch, D rows prefetched>
//   lane `tid`, unit j  <->  mat columns 2*(tid + j*T) + {0,1}  <->  reference columns +1
//   workgroup b owns rows b, b+NB, b+2NB, ...; lanes 0..R-1 of wave 0 also own the scalar side
//   (RHS entry, pivot-column entry, ratio) of row g = lane.
// Load discipline: every load is unconditional with an in-bounds (possibly dummy) address and
// the body is fully unrolled, so hipcc can count the load queue (s_waitcnt vmcnt(N)) instead of
// draining it; vmcnt retires in issue order, hence the issue order below is deliberate.
// ------------------------------------------------------------------------------------------
template <int T, int J, int R, int D>
__global__ __launch_bounds__(T) void pivot_kernel(Desc d, int parity, int mode, int force, const double *gather) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_la[2][R]; // next entering column's entries of my rows (ping-pong per batch)
    __shared__ int cyc_flag;

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING || Sin->pause) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int64_t pivots_in = Sin->pivots, hist_len_in = Sin->hist_len;
    // (0) pending basis bookkeeping of the previous pivot: its two loads are the oldest of this
    // launch, its four stores are fire-and-forget at the end (or before any early return)
    const bool swapper = b == 0 && tid == 0 && Sin->swap_valid;
    const int sw_row = Sin->swap_row, sw_col = Sin->swap_col;
    int sw_leaving = 0, sw_entering = 0;
    bool swapped = false;
    if (swapper) {
        sw_leaving = d.var[d.w + sw_row];
        sw_entering = d.var[sw_col];
    }
    auto apply_swap = [&]() {
        if (swapper && !swapped) {
            d.var[d.w + sw_row] = sw_entering;
            d.var[sw_col] = sw_leaving;
            d.pos[sw_leaving] = sw_col;
            d.pos[sw_entering] = d.w + sw_row;
        }
        swapped = true;
    };
    // every field of the next state, from registers
    auto write_state = [&](int status, int phase_, int bootstrap_, int la_, int pbuf_, int mbuf_, int pause_,
                           int dec_valid_, int dec_row_, int dec_col_, int swap_valid_, int swap_row_,
                           int swap_col_, int64_t hist_len_, double iter_, double result_, int64_t pivots_) {
        Sout->status = status;
        Sout->phase = phase_;
        Sout->bootstrap = bootstrap_;
        Sout->la = la_;
        Sout->pbuf = pbuf_;
        Sout->mbuf = mbuf_;
        Sout->pause = pause_;
        Sout->dec_valid = dec_valid_;
        Sout->dec_row = dec_row_;
        Sout->dec_col = dec_col_;
        Sout->swap_valid = swap_valid_;
        Sout->swap_row = swap_row_;
        Sout->swap_col = swap_col_;
        Sout->pad_ = 0;
        Sout->hist_len = hist_len_;
        Sout->iter = iter_;
        Sout->result = result_;
        Sout->pivots = pivots_;
    };
    // Every launch that gets past the selection flips both ping-pong indices, so in FUSED graphs
    // they equal the launch parity (a kernel argument): the first loads need not wait for the state.
    const int pbuf = mode == MODE_FUSED ? parity : Sin->pbuf;
    const int mbuf = mode == MODE_FUSED ? parity : Sin->mbuf;
    const int la_in = Sin->la;
    const double *__restrict__ matA = d.mat[mbuf];
    const double *__restrict__ rhsA = d.rhs[mbuf];
    double *__restrict__ matB = d.mat[mbuf ^ 1];
    double *__restrict__ rhsB = d.rhs[mbuf ^ 1];
    const bool bootstrap = Sin->bootstrap != 0;
    const int phase_in = Sin->phase;
    const double iter_in = Sin->iter;
    int phase = phase_in;
    double iter = iter_in;
    bool phase_switched = false;
    int slot = 0; // block_argmin scratch ping-pong

    // lane's column offsets (lanes past the row end use column 0: in-bounds dummy)
    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }

    // (1) control loads first (oldest in the queue): partials of the previous launch, objective row
    // (SHARD mode: the all-gathered per-rank candidates instead -- slot layout at shard_select_kernel)
    const int gstride = SHARD_HDR + 2 * pitch;
    const int ncand = mode == MODE_SHARD ? d.nshards : NB;
    Part p_rhs, p_ratio;
    if (mode == MODE_SHARD) {
        const double *slot_ = gather + (size_t)(tid < ncand ? tid : 0) * gstride;
        p_ratio.key = slot_[0];
        p_ratio.idx = (int)slot_[1];
        p_rhs.key = slot_[2];
        p_rhs.idx = (int)slot_[3];
    } else {
        const int pi = tid < NB ? tid : 0;
        p_rhs = d.part_rhs[pbuf][pi];
        p_ratio = d.part_ratio[pbuf][pi];
    }
    // where a (global) row's raw data and RHS entry come from: my tableau, or its owner's gather slot
    auto owner_slot = [&](int grow) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < MAX_SHARDS; k++)
            if (k < d.nshards && grow >= d.bounds[k]) g = k;
        return gather + (size_t)g * gstride;
    };
    double2 o[J]; // objective row slice (reduced costs)
#pragma unroll
    for (int j = 0; j < J; j++) o[j] = *reinterpret_cast<const double2 *>(matA + cofs[j]);

    // (2) the first D rows of my first batch: they depend on nothing the selection decides, so
    // they stream in while the selection runs
    double2 x[R][J];
    {
#pragma unroll
        for (int g = 0; g < D; g++) {
            const int r = b + NB * g;
            const double *mr = matA + (size_t)(r < h ? r : b) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) x[g][j] = ld_row(mr + cofs[j], false);
        }
    }

    int row = 0, col = 0;
    bool have_pivot = false, pv_loaded = false;
    double2 pv[J]; // pivot row slice: raw, then normalised
#pragma unroll
    for (int j = 0; j < J; j++) pv[j] = make_double2(0.0, 0.0);

    // ---------------- decide: which pivot, or stop (src/simplex.ts:66-142 minus pivot()) ------
    if (mode != MODE_APPLY && !bootstrap) {
        int term = RUNNING;
        double term_result = NAN;
        for (;;) {
            if (!(iter < max_pivots)) { // loop bounds :69,109 -> "cycled" :102,141
                term = YALPS_CYCLED;
                break;
            }
            if (phase == 1) {
                // leaving row: most negative RHS, strict <, first wins (:111-119)
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_rhs.key;
                    c.i = p_rhs.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) { // :120 tail call of phase2: fresh counter and history
                    phase = 2;
                    iter = 0.0;
                    phase_switched = true;
                    continue;
                }
                row = c.i;
                // entering column: max -M[0,c]/M[row,c] over M[row,c] < -precision (:123-134)
                const double *mrow = mode == MODE_SHARD ? owner_slot(row) + SHARD_HDR + pitch
                                                       : matA + (size_t)row * pitch;
                KI e = {INFINITY, INT_MAX};
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    pv[j] = *reinterpret_cast<const double2 *>(mrow + cofs[j]);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const double coefficient = elem(pv[j], k);
                        if (c0 + k < n && coefficient < -precision) {
                            const double ratio = -elem(o[j], k) / coefficient;
                            if (ratio > -INFINITY && ki_better(-ratio, c0 + k + 1, e.k, e.i)) {
                                e.k = -ratio;
                                e.i = c0 + k + 1;
                            }
                        }
                    }
                }
                pv_loaded = true;
                e = block_argmin<T>(e, sk, si, slot);
                slot ^= 1;
                if (e.i == INT_MAX) { // :135
                    term = YALPS_INFEASIBLE;
                    break;
                }
                col = e.i;
                break;
            } else {
                col = la_in; // Dantzig pricing (:71-79) was done by the previous launch
                if (col == 0) { // :80
                    term = YALPS_OPTIMAL;
                    term_result = round_to_precision(rhsA[0], precision);
                    break;
                }
                // leaving row: min-ratio test with the early break (:83-95); the partials carry
                // key = -inf for "ratio <= precision" rows so the lowest such index wins
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
        int64_t hist_len = phase_switched ? 0 : hist_len_in;
        if (term != RUNNING) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(term, phase, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len, iter, term_result,
                            pivots_in);
            return;
        }
        if (C->check_cycles) { // :98,137 (DECIDE launches only: one workgroup)
            if (hist_len >= C->hist_cap) { // history full: the host grows it; nothing is consumed
                apply_swap();
                if (tid == 0)
                    write_state(RUNNING, phase, 0, la_in, pbuf, mbuf, 1, 0, 0, 0, 0, 0, 0, hist_len, iter, NAN,
                                pivots_in);
                return;
            }
            apply_swap(); // the detector reads the basis as it is now
            __syncthreads();
            const bool cyc = has_cycle(C, hist_len, d.var[d.w + row], d.var[col], &cyc_flag);
            hist_len += 1;
            if (cyc) {
                if (tid == 0)
                    write_state(YALPS_CYCLED, phase, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len, iter, NAN,
                                pivots_in);
                return;
            }
        }
        have_pivot = true;
        if (mode == MODE_DECIDE) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(RUNNING, phase, 0, la_in, pbuf, mbuf, 0, 1, row, col, 0, 0, 0, hist_len, iter + 1.0, NAN,
                            pivots_in + 1);
            return;
        }
    } else if (mode == MODE_APPLY && Sin->dec_valid) {
        row = Sin->dec_row;
        col = Sin->dec_col;
        have_pivot = true;
    }
    if (!have_pivot && !bootstrap) { // APPLY with nothing decided
        apply_swap();
        if (b == 0 && tid == 0 && !(force & 1))
            write_state(RUNNING, phase_in, 0, la_in, pbuf, mbuf, 0, 0, 0, 0, 0, 0, 0, hist_len_in, iter_in, NAN,
                        pivots_in);
        return;
    }

    // ---------------- prepare: pivot row normalise + look-ahead pricing -----------------------
    // owner lane/unit/element of a reference column c (c >= 1): mat column c-1
    const int ucol = (col - 1) >> 1, ecol = (col - 1) & 1;
    const int col_tid = have_pivot ? ucol % T : -1, col_j = have_pivot ? ucol / T : -1;
    const int colx = have_pivot ? col - 1 : 0; // in-bounds even without a pivot
    // (3) pivot row, quotient, objective row's pivot-column entry
    // (SHARD: the pivot row travels in its owner's gather slot -- the ratio candidate's row in
    // phase 2, the most-negative-RHS candidate's row in phase 1; `row` is a GLOBAL index and
    // `lrow` its local index here, -1 if another rank owns it)
    const double *gslot = mode == MODE_SHARD ? owner_slot(row) : nullptr;
    const double *mrow = mode == MODE_SHARD ? gslot + SHARD_HDR + (phase == 1 ? pitch : 0) : matA + (size_t)row * pitch;
    const int lrow = !have_pivot ? -1
                     : mode != MODE_SHARD ? row
                     : (row >= d.bounds[d.shard_rank] && row < d.bounds[d.shard_rank + 1]) ? row - d.row_base : -1;
    if (!pv_loaded) {
#pragma unroll
        for (int j = 0; j < J; j++) pv[j] = *reinterpret_cast<const double2 *>(mrow + cofs[j]);
    }
    const double q_ld = mrow[colx], coef0_ld = matA[colx];
    const double rhs_row = mode == MODE_SHARD ? gslot[phase == 1 ? 5 : 4] : rhsA[row];
    const double q = have_pivot ? q_ld : 1.0, coef0 = have_pivot ? coef0_ld : 0.0;

    unsigned nzmask = 0; // bit (2j+k): pivot-row entry is in nonZeroColumns (:18-23)
    if (have_pivot) {    // src/simplex.ts:14-25
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double v = elem(pv[j], k);
                const bool nz = fabs(v) > 1e-16;
                pv[j] = with_elem(pv[j], k, nz ? v / q : 0.0);
                if (nz) nzmask |= 1u << (2 * j + k);
            }
        }
    }
    const double inv_q = 1.0 / q; // :25 (the pivot entry becomes 1/quotient)
    int la = 0, la_tid = -1, la_j = -1, ela = 0;
    bool la_known = false;

    // ---------------- body: eliminate my rows into the other buffer, emit partials ------------
    KI cand_ratio = {INFINITY, INT_MAX}, cand_rhs = {INFINITY, INT_MAX}; // lanes 0..R-1
    for (int i0 = 0; b + NB * i0 < h; i0 += R) {
        const int r_first = b + NB * i0;
        // (4) pivot-column entries of my rows (uniform per row) and, lane g, the RHS of row g
        double coef[R];
#pragma unroll
        for (int g = 0; g < R; g++) {
            const int r = b + NB * (i0 + g);
            coef[g] = matA[(size_t)(r < h ? r : r_first) * pitch + colx];
        }
        const int my_r = b + NB * (i0 + tid);
        const bool my_live = tid < R && my_r < h;
        const double rr = rhsA[my_live ? my_r : 0];
        if (i0 > 0) { // later batches: prefetch their first D rows (batch 0's came in at the top)
#pragma unroll
            for (int g = 0; g < D; g++) {
                const int r = b + NB * (i0 + g);
                const double *mr = matA + (size_t)(r < h ? r : r_first) * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) x[g][j] = ld_row(mr + cofs[j], false);
            }
        }
        // lane g: scalar side of row g -- RHS entry (:33 at c = 0) and pivot-column entry (:36)
        double my_rhs = rr, my_val = 0.0;
        bool my_val_set = false;
        if (my_live && have_pivot) {
            double my_coef = 0.0;
#pragma unroll
            for (int g = 0; g < R; g++)
                if (tid == g) my_coef = coef[g];
            const bool nz_rhs = fabs(rhs_row) > 1e-16;
            const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
            if (my_r == lrow) {
                my_rhs = pn_rhs;
            } else if (fabs(my_coef) > 1e-16) {
                if (nz_rhs) {
                    const double prod = my_coef * pn_rhs;
                    my_rhs = rr - prod;
                }
                const double nq = -my_coef / q;
                matB[(size_t)my_r * pitch + colx] = nq; // the owner lane stores only the other half
                my_val = nq; // what this row holds in column `col` from now on
                my_val_set = true;
            }
            rhsB[my_r] = my_rhs;
        } else if (my_live) {
            rhsB[my_r] = rr; // bootstrap: carry over
        }
        // rows: eliminate + write to the other buffer, row by row, with the loads of the next
        // rows in flight.  Rows the reference leaves untouched (:31) are carried over unchanged.
#pragma unroll
        for (int g = 0; g < R; g++) {
            const int r = b + NB * (i0 + g);
            const bool live = r < h;
            const double c = coef[g];
            const bool act = have_pivot && live && r != lrow && fabs(c) > 1e-16;
            if (have_pivot && live && r == lrow) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    x[g][j] = pv[j];
                    if (tid == col_tid && j == col_j) x[g][j] = with_elem(x[g][j], ecol, inv_q);
                }
            } else if (act) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    if (nzmask & (1u << (2 * j))) {
                        const double prod = c * pv[j].x;
                        x[g][j].x = x[g][j].x - prod;
                    }
                    if (nzmask & (1u << (2 * j + 1))) {
                        const double prod = c * pv[j].y;
                        x[g][j].y = x[g][j].y - prod;
                    }
                }
            }
            if (live) { // (a bootstrap launch just carries the tableau over)
                double *mr = matB + (size_t)r * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 >= pitch) continue;
                    if (act && tid == col_tid && j == col_j) // lane g of wave 0 stores -coef/quotient
                        mr[c0 + (ecol ^ 1)] = elem(x[g][j], ecol ^ 1);
                    else if (force & 64) { // streaming (non-temporal) stores; YALPS_HIP_NT=0 turns them off
                        __builtin_nontemporal_store(x[g][j].x, mr + c0);
                        __builtin_nontemporal_store(x[g][j].y, mr + c0 + 1);
                    } else
                        *reinterpret_cast<double2 *>(mr + c0) = x[g][j];
                }
            }
            if (g + D < R) { // keep D rows in flight
                const int rn = b + NB * (i0 + g + D);
                const double *mr = matA + (size_t)(rn < h ? rn : r_first) * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) x[g + D][j] = ld_row(mr + cofs[j], false);
            }
        }
        // Dantzig pricing (:71-79) of the objective row as it is AFTER this pivot (as it is, when
        // bootstrapping) -> entering column `la` of the next iteration
        if (!la_known) {
            la_known = true;
            const bool touched0 = have_pivot && fabs(coef0) > 1e-16;
            KI best = {INFINITY, INT_MAX};
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    double ov = elem(o[j], k);
                    if (touched0) {
                        if (tid == col_tid && j == col_j && k == ecol)
                            ov = -coef0 / q;
                        else if (nzmask & (1u << (2 * j + k))) {
                            const double prod = coef0 * elem(pv[j], k);
                            ov = ov - prod;
                        }
                    }
                    if (c0 + k < n && ov > precision && ki_better(-ov, c0 + k + 1, best.k, best.i)) {
                        best.k = -ov;
                        best.i = c0 + k + 1;
                    }
                }
            }
            best = block_argmin<T>(best, sk, si, slot);
            slot ^= 1;
            la = best.i == INT_MAX ? 0 : best.i;
            const int ula = (la - 1) >> 1;
            ela = (la - 1) & 1;
            la_tid = la > 0 ? ula % T : -1;
            la_j = la > 0 ? ula / T : -1;
        }
        // entries of my rows in column `la` (x holds the rows as written), for lanes 0..R-1
        if (tid == la_tid) {
#pragma unroll
            for (int g = 0; g < R; g++)
#pragma unroll
                for (int j = 0; j < J; j++)
                    if (j == la_j) sh_la[(i0 / R) & 1][g] = elem(x[g][j], ela);
        }
        __syncthreads(); // sh_la of this batch visible to lanes 0..R-1
        // candidates of my row for the next launch's scans
        if (my_live && my_r >= 1) {
            const int my_gr = my_r + d.row_base; // global row index
            if (my_rhs < -precision && ki_better(my_rhs, my_gr, cand_rhs.k, cand_rhs.i)) {
                cand_rhs.k = my_rhs;
                cand_rhs.i = my_gr;
            }
            if (la > 0) {
                const double value = (my_val_set && la == col) ? my_val : sh_la[(i0 / R) & 1][tid];
                if (value > precision) {
                    const double ratio = my_rhs / value;
                    if (ratio < INFINITY) {
                        const double key = (ratio <= precision) ? -INFINITY : ratio;
                        if (ki_better(key, my_gr, cand_ratio.k, cand_ratio.i)) {
                            cand_ratio.k = key;
                            cand_ratio.i = my_gr;
                        }
                    }
                }
            }
        }
    }
    if (tid < 64) { // R <= 16 < 64: wave 0 holds every candidate
        cand_ratio = wave_argmin(cand_ratio);
        cand_rhs = wave_argmin(cand_rhs);
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
    }
    apply_swap();
    if (b == 0 && tid == 0 && !(force & 1)) {
        const bool counted = have_pivot && mode != MODE_APPLY; // DECIDE already counted an APPLY's pivot
        write_state(RUNNING, phase, 0, la, pbuf ^ 1, mbuf ^ 1, 0, 0, 0, 0, have_pivot ? 1 : 0, row, col,
                    (mode != MODE_APPLY && phase_switched) ? 0 : hist_len_in, counted ? iter + 1.0 : iter, NAN,
                    counted ? pivots_in + 1 : pivots_in);
    }
}

// ------------------------------------------------------------------------------------------
// wide_kernel<T lanes, J units per lane per row>: the streaming pivot for tableaux whose rows are
// too wide (or whose workgroups own too many rows) to keep a batch of rows plus the pivot row and
// the objective row in registers (n > 4096 columns: 16385-wide row shards; or > 16 rows per
// workgroup: 4097^2).  Same launch protocol, state, partials and modes (FUSED / APPLY / SHARD)
// as pivot_kernel; the differences are the data flow:
//   * the normalised pivot row lives in LDS (8 B per column, <= 131 KB; FLUSHED marks the
//     entries pivot() zeroed) and is read back 16 B per lane and row (LDS rate >> HBM rate);
//   * the objective row is streamed (twice at most) instead of held;
//   * my rows are streamed ONE at a time, double-buffered: the loads of row i+1 are in flight
//     while row i is eliminated and stored -- register use is independent of the row count;
//   * the scalar side of every row (RHS entry, -coef/quotient) is computed uniformly by all
//     lanes; the rows' entries of the next entering column and their new RHS are parked in LDS
//     and turned into this workgroup's partial after the last row.
// ------------------------------------------------------------------------------------------
template <int T, int J>
__global__ __launch_bounds__(T) void wide_kernel(Desc d, int parity, int mode, int force, const double *gather) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double wd_dyn[]; // prow[pitch], lav[rpw], rhsv[rpw]

    const int tid = threadIdx.x, NB = d.nb, b = blockIdx.x;
    const YState *Sin = d.st + parity;
    YState *Sout = d.st + (parity ^ 1);
    const YConst *C = d.cst;
    if (Sin->status != RUNNING || Sin->pause) {
        if (b == 0 && tid == 0) state_copy(Sout, Sin);
        return;
    }
    const int h = C->height, n = d.n, pitch = d.pitch;
    const int rpw = (d.hcap + NB - 1) / NB; // rows per workgroup (capacity)
    double *prow = wd_dyn, *lav = wd_dyn + pitch, *rhsv = lav + rpw;
    const double precision = C->precision, max_pivots = C->max_pivots;
    const int64_t pivots_in = Sin->pivots, hist_len_in = Sin->hist_len;
    const bool swapper = b == 0 && tid == 0 && Sin->swap_valid;
    const int sw_row = Sin->swap_row, sw_col = Sin->swap_col;
    int sw_leaving = 0, sw_entering = 0;
    bool swapped = false;
    if (swapper) {
        sw_leaving = d.var[d.w + sw_row];
        sw_entering = d.var[sw_col];
    }
    auto apply_swap = [&]() __attribute__((always_inline)) {
        if (swapper && !swapped) {
            d.var[d.w + sw_row] = sw_entering;
            d.var[sw_col] = sw_leaving;
            d.pos[sw_leaving] = sw_col;
            d.pos[sw_entering] = d.w + sw_row;
        }
        swapped = true;
    };
    auto write_state = [&](int status, int phase_, int la_, int pbuf_, int mbuf_, int swap_valid_, int swap_row_,
                           int swap_col_, int64_t hist_len_, double iter_, double result_, int64_t pivots_)
                           __attribute__((always_inline)) {
        Sout->status = status;
        Sout->phase = phase_;
        Sout->bootstrap = 0;
        Sout->la = la_;
        Sout->pbuf = pbuf_;
        Sout->mbuf = mbuf_;
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
    const int pbuf = mode == MODE_FUSED ? parity : Sin->pbuf;
    const int mbuf = mode == MODE_FUSED ? parity : Sin->mbuf;
    const int la_in = Sin->la;
    const double *__restrict__ matA = d.mat[mbuf];
    const double *__restrict__ rhsA = d.rhs[mbuf];
    double *__restrict__ matB = d.mat[mbuf ^ 1];
    double *__restrict__ rhsB = d.rhs[mbuf ^ 1];
    const bool bootstrap = Sin->bootstrap != 0;
    const int phase_in = Sin->phase;
    const double iter_in = Sin->iter;
    int phase = phase_in;
    double iter = iter_in;
    bool phase_switched = false;
    int slot = 0;
    const int units = pitch / 2;

    const int gstride = SHARD_HDR + 2 * pitch;
    const int ncand = mode == MODE_SHARD ? d.nshards : NB;
    Part p_rhs, p_ratio;
    if (mode == MODE_SHARD) {
        const double *slot_ = gather + (size_t)(tid < ncand ? tid : 0) * gstride;
        p_ratio.key = slot_[0];
        p_ratio.idx = (int)slot_[1];
        p_rhs.key = slot_[2];
        p_rhs.idx = (int)slot_[3];
    } else {
        const int pi = tid < NB ? tid : 0;
        p_rhs = d.part_rhs[pbuf][pi];
        p_ratio = d.part_ratio[pbuf][pi];
    }
    auto owner_slot = [&](int grow) __attribute__((always_inline)) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < MAX_SHARDS; k++)
            if (k < d.nshards && grow >= d.bounds[k]) g = k;
        return gather + (size_t)g * gstride;
    };

    // ---------------- decide ------------------------------------------------------------------
    int row = 0, col = 0;
    bool have_pivot = false;
    if (mode != MODE_APPLY && !bootstrap) {
        int term = RUNNING;
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
                if (c.i == INT_MAX) {
                    phase = 2;
                    iter = 0.0;
                    phase_switched = true;
                    continue;
                }
                row = c.i;
                const double *mrow1 = mode == MODE_SHARD ? owner_slot(row) + SHARD_HDR + pitch : matA + (size_t)row * pitch;
                KI e = {INFINITY, INT_MAX};
                for (int cc = tid; cc < n; cc += T) { // src/simplex.ts:123-134
                    const double coefficient = mrow1[cc];
                    if (coefficient < -precision) {
                        const double ratio = -matA[cc] / coefficient;
                        if (ratio > -INFINITY && ki_better(-ratio, cc + 1, e.k, e.i)) {
                            e.k = -ratio;
                            e.i = cc + 1;
                        }
                    }
                }
                e = block_argmin<T>(e, sk, si, slot);
                slot ^= 1;
                if (e.i == INT_MAX) {
                    term = YALPS_INFEASIBLE;
                    break;
                }
                col = e.i;
                break;
            } else {
                col = la_in;
                if (col == 0) {
                    term = YALPS_OPTIMAL;
                    term_result = round_to_precision(rhsA[0], precision);
                    break;
                }
                KI c = {INFINITY, INT_MAX};
                if (tid < ncand) {
                    c.k = p_ratio.key;
                    c.i = p_ratio.idx;
                }
                c = block_argmin<T>(c, sk, si, slot);
                slot ^= 1;
                if (c.i == INT_MAX) {
                    term = YALPS_UNBOUNDED;
                    term_result = (double)col;
                    break;
                }
                row = c.i;
                break;
            }
        }
        if (term != RUNNING) {
            apply_swap();
            if (b == 0 && tid == 0)
                write_state(term, phase, la_in, pbuf, mbuf, 0, 0, 0, phase_switched ? 0 : hist_len_in, iter, term_result,
                            pivots_in);
            return;
        }
        have_pivot = true;
    } else if (mode == MODE_APPLY && Sin->dec_valid) {
        row = Sin->dec_row;
        col = Sin->dec_col;
        have_pivot = true;
    }
    if (!have_pivot && !bootstrap) {
        apply_swap();
        if (b == 0 && tid == 0 && !(force & 1))
            write_state(RUNNING, phase_in, la_in, pbuf, mbuf, 0, 0, 0, hist_len_in, iter_in, NAN, pivots_in);
        return;
    }

    // ---------------- prepare: normalised pivot row -> LDS, look-ahead pricing --------------------
    const int colx = have_pivot ? col - 1 : 0;
    const double *gslot = mode == MODE_SHARD ? owner_slot(row) : nullptr;
    const double *mrow = mode == MODE_SHARD ? gslot + SHARD_HDR + (phase == 1 ? pitch : 0) : matA + (size_t)row * pitch;
    const int lrow = !have_pivot ? -1
                     : mode != MODE_SHARD ? row
                     : (row >= d.bounds[d.shard_rank] && row < d.bounds[d.shard_rank + 1]) ? row - d.row_base : -1;
    const double q = have_pivot ? mrow[colx] : 1.0;
    const double coef0 = have_pivot ? matA[colx] : 0.0;
    const double rhs_row = have_pivot ? (mode == MODE_SHARD ? gslot[phase == 1 ? 5 : 4] : rhsA[row]) : 0.0;
    const double inv_q = 1.0 / q;
    const double flushed = __longlong_as_double((long long)FLUSHED);
    for (int u = tid; u < units; u += T) {
        double2 v = have_pivot ? *reinterpret_cast<const double2 *>(mrow + 2 * u) : make_double2(0.0, 0.0);
        v.x = fabs(v.x) > 1e-16 ? v.x / q : flushed;
        v.y = fabs(v.y) > 1e-16 ? v.y / q : flushed;
        *reinterpret_cast<double2 *>(prow + 2 * u) = v;
    }
    __syncthreads();
    int la = 0;
    {
        const bool touched0 = have_pivot && fabs(coef0) > 1e-16;
        KI best = {INFINITY, INT_MAX};
        for (int cc = tid; cc < n; cc += T) {
            double ov = matA[cc];
            if (touched0) {
                const double pn = prow[cc];
                if (cc == colx)
                    ov = -coef0 / q;
                else if ((unsigned long long)__double_as_longlong(pn) != FLUSHED) {
                    const double prod = coef0 * pn;
                    ov = ov - prod;
                }
            }
            if (ov > precision && ki_better(-ov, cc + 1, best.k, best.i)) {
                best.k = -ov;
                best.i = cc + 1;
            }
        }
        best = block_argmin<T>(best, sk, si, slot);
        slot ^= 1;
        la = best.i == INT_MAX ? 0 : best.i;
    }
    const int lax = la > 0 ? la - 1 : -1; // mat column of the next entering variable
    const bool nz_rhs = fabs(rhs_row) > 1e-16;
    const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;

    // ---------------- body: stream my rows, one at a time, double-buffered ----------------------
    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    const int my_rows = b < h ? (h - 1 - b) / NB + 1 : 0;
    auto load_row = [&](double2 (&x)[J], double &cf, double &rr, int i) __attribute__((always_inline)) {
        const int r = b + NB * i;
        const int rs = r < h ? r : b; // in-bounds dummy past the end (b < h whenever this is reached)
        cf = matA[(size_t)rs * pitch + colx];
        rr = rhsA[rs];
        const double *mr = matA + (size_t)rs * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) x[j] = *reinterpret_cast<const double2 *>(mr + cofs[j]);
    };
    auto process = [&](double2 (&x)[J], double cf, double rr, int i) __attribute__((always_inline)) {
        const int r = b + NB * i;
        if (r >= h) return;
        double my_rhs = rr;
        const bool is_pivot_row = have_pivot && r == lrow;
        const bool act = have_pivot && !is_pivot_row && fabs(cf) > 1e-16; // src/simplex.ts:31
        const double nq = act ? -cf / q : 0.0;                            // :36 (uniform)
        if (is_pivot_row)
            my_rhs = pn_rhs;
        else if (act && nz_rhs) {
            const double prod = cf * pn_rhs;
            my_rhs = rr - prod;
        }
        double *mr = matB + (size_t)r * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (c0 >= pitch) continue;
            double2 v = x[j];
            if (is_pivot_row || act) {
                const double2 pn = *reinterpret_cast<const double2 *>(prow + c0);
                const bool f0 = (unsigned long long)__double_as_longlong(pn.x) != FLUSHED;
                const bool f1 = (unsigned long long)__double_as_longlong(pn.y) != FLUSHED;
                if (is_pivot_row) {
                    v.x = f0 ? pn.x : 0.0;
                    v.y = f1 ? pn.y : 0.0;
                    if (c0 == colx) v.x = inv_q; // :25
                    if (c0 + 1 == colx) v.y = inv_q;
                } else {
                    const double px = cf * pn.x, py = cf * pn.y;
                    const double nx = v.x - px, ny = v.y - py;
                    v.x = f0 ? nx : v.x;
                    v.y = f1 ? ny : v.y;
                    if (c0 == colx) v.x = nq;
                    if (c0 + 1 == colx) v.y = nq;
                }
            }
            if (c0 == (lax & ~1)) lav[i] = (lax & 1) ? v.y : v.x; // the row's entry in the next entering column
            if (force & 64) {
                __builtin_nontemporal_store(v.x, mr + c0);
                __builtin_nontemporal_store(v.y, mr + c0 + 1);
            } else
                *reinterpret_cast<double2 *>(mr + c0) = v;
        }
        if (tid == 0) {
            rhsB[r] = my_rhs;
            rhsv[i] = my_rhs;
        }
    };
    if (my_rows > 0) {
        double2 xa[J], xb[J];
        double cfa, cfb, rra, rrb;
        load_row(xa, cfa, rra, 0);
        for (int i = 0; i < my_rows; i += 2) {
            load_row(xb, cfb, rrb, i + 1);
            process(xa, cfa, rra, i);
            load_row(xa, cfa, rra, i + 2);
            process(xb, cfb, rrb, i + 1);
        }
    }
    __syncthreads(); // lav / rhsv complete
    KI cand_ratio = {INFINITY, INT_MAX}, cand_rhs = {INFINITY, INT_MAX};
    for (int i = tid; i < my_rows; i += T) {
        const int r = b + NB * i;
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
    apply_swap();
    if (b == 0 && tid == 0 && !(force & 1)) {
        const bool counted = have_pivot && mode != MODE_APPLY;
        write_state(RUNNING, phase, la, pbuf ^ 1, mbuf ^ 1, have_pivot ? 1 : 0, row, col,
                    (mode != MODE_APPLY && phase_switched) ? 0 : hist_len_in, counted ? iter + 1.0 : iter, NAN,
                    counted ? pivots_in + 1 : pivots_in);
    }
}

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
    if (tid == 0) {
        send[0] = cr.k;
        send[1] = (double)cr.i;
        send[2] = cn.k;
        send[3] = (double)cn.i;
        send[4] = rhs[lr];
        send[5] = rhs[ln];
        send[6] = 0.0;
        send[7] = 0.0;
    }
    for (int c = tid; c < pitch; c += 1024) {
        send[SHARD_HDR + c] = mat[(size_t)lr * pitch + c];
        send[SHARD_HDR + pitch + c] = mat[(size_t)ln * pitch + c];
    }
}

// ------------------------------------------------------------------------------------------
// resident_kernel: the whole pivot loop in ONE launch, tableau resident in the register files.
//
// Applies when the tableau fits on chip (2049 x 2049 fp64 = 33.6 MB against 128 MB of VGPRs): one
// workgroup per CU keeps its rows (b, b+NB, ...) in registers for the whole solve; every workgroup
// also keeps a replica of the objective row.  Per pivot the ONLY traffic is one exchange through
// L2: every workgroup publishes its candidate (min-ratio row in phase 2, most-negative-RHS row in
// phase 1) together with that row's data, all workgroups read the NB (key, row) pairs, take the
// same arg-min and fetch the winner's row.  Nothing is streamed from or to HBM inside the loop.
//
// Hand-off = Guideline 16 R1 of the CDNA guide, table row 1: payload stored write-through (agent-
// scope relaxed atomic stores = sc1), every storing wave drains (s_waitcnt vmcnt(0)), workgroup
// barrier, ONE lane stores the flag {epoch, row}; consumers poll that one word per producer with
// sc1 loads, join a workgroup barrier, then read the payload with sc1 loads only.  Buffers are
// ping-ponged by epoch parity: a workgroup cannot get two epochs ahead of another one because it
// needs that workgroup's flag of the epoch in between.  Results do not depend on placement or
// timing: every decision is a deterministic function of bytes that are identical for all readers.
// Every spin is bounded; a give-up sets rc_err and the host re-runs the chunk with the streaming
// kernel from the untouched input buffer.
//
// The kernel runs at most `chunk` pivots per launch (bounded run time; the host relaunches while
// the status is RUNNING) and writes the tableau to the OTHER buffer on exit.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_sc1(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// 16-byte forms (one instruction per lane unit).  hipcc does not count inline-asm memory operations:
// the store is covered by the publisher's explicit s_waitcnt vmcnt(0), the load waits inside its
// own statement (CDNA guide 5.7, form (i)).
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16_sc1(double *p, double2 v) {
    v2f64 t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
template <int J>
__device__ __forceinline__ void ld16_sc1(double2 (&out)[J], const double *base, const int (&ofs)[J]) {
    static_assert(J == 1 || J == 2 || J == 4, "lane units per row");
    v2f64 t[J];
    if constexpr (J == 1) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t[0]) : "v"(base + ofs[0]) : "memory");
    } else if constexpr (J == 2) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1])
                     : "v"(base + ofs[0]), "v"(base + ofs[1])
                     : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2]), "v"(base + ofs[3])
                     : "memory");
    }
#pragma unroll
    for (int j = 0; j < J; j++) out[j] = make_double2(t[j].x, t[j].y);
}

template <int T, int J, int R>
__global__ __launch_bounds__(T) void resident_kernel(Desc d, int parity, int chunk) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_val[R + 2]; // per-row broadcast: pivot-column entry / entering-column entry
    __shared__ double sh_nq[R + 2];  // -coef/quotient per row (:36), for the objective row, 1/quotient (:25)
    __shared__ double sh_ck;         // my candidate for the next exchange: key, row, local slot
    __shared__ int sh_ci, sh_cg, sh_fail;
    extern __shared__ int sh_perm[]; // workgroup 0: var[perm_len] then pos[perm_len]

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
    const double *matA = d.mat[mbuf];
    const double *rhsA = d.rhs[mbuf];
    int phase = Sin->phase;
    double iter = Sin->iter;
    int64_t pivots = Sin->pivots;
    int slot = 0;

    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    // ---- load my rows, the objective replica, my rows' RHS (lane g), the basis (workgroup 0) ----
    double2 x[R][J], o[J];
#pragma unroll
    for (int j = 0; j < J; j++) o[j] = *reinterpret_cast<const double2 *>(matA + cofs[j]);
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int r = b + NB * g;
        const double *mr = matA + (size_t)(r < h ? r : b) * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) x[g][j] = *reinterpret_cast<const double2 *>(mr + cofs[j]);
    }
    const int my_r = b + NB * tid; // lane g = tid < R owns the scalar side of row g
    const bool my_live = tid < R && my_r < h;
    double my_rhs = rhsA[my_live ? my_r : 0];
    if (b == 0) {
        for (int i = tid; i < d.perm_len; i += T) {
            sh_perm[i] = d.var[i];
            sh_perm[d.perm_len + i] = d.pos[i];
        }
    }
    if (tid == 0) sh_fail = 0;
    __syncthreads();

    // ---- building blocks of one round ------------------------------------------------------------
    int la = 0; // entering column of the NEXT pivot (phase 2), priced on my objective replica
    // Dantzig pricing (src/simplex.ts:71-79) on my replica of the objective row -> la
    auto price = [&]() __attribute__((always_inline)) {
        KI best = {INFINITY, INT_MAX};
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double ov = elem(o[j], k);
                if (c0 + k < n && ov > precision && ki_better(-ov, c0 + k + 1, best.k, best.i)) {
                    best.k = -ov;
                    best.i = c0 + k + 1;
                }
            }
        }
        best = block_argmin<T>(best, sk, si, slot);
        slot ^= 1;
        la = best.i == INT_MAX ? 0 : best.i;
    };
    // lanes 0..R-1: candidate of my row of the given kind (1 = most negative RHS, 2 = min ratio with
    // the row's entry in column la taken from sh_val[lane]), reduced over the workgroup and left in
    // sh_ck / sh_ci / sh_cg for every lane
    auto candidate = [&](int kind) __attribute__((always_inline)) {
        KI c = {INFINITY, INT_MAX};
        if (my_live && my_r >= 1) {
            if (kind == 1) {
                if (my_rhs < -precision) {
                    c.k = my_rhs;
                    c.i = my_r;
                }
            } else if (la > 0) {
                const double value = sh_val[tid];
                if (value > precision) {
                    const double ratio = my_rhs / value;
                    if (ratio < INFINITY) {
                        c.k = (ratio <= precision) ? -INFINITY : ratio;
                        c.i = my_r;
                    }
                }
            }
        }
        if (tid < 64) {
            c = wave_argmin(c);
            if (tid == 0) {
                sh_ck = c.k;
                sh_ci = c.i;
                sh_cg = c.i == INT_MAX ? 0 : c.i / NB;
            }
        }
        __syncthreads();
    };
    unsigned epoch = 0;
    // publish my candidate (sh_ck / sh_ci) and the data of its row (register slot sh_cg)
    auto publish = [&]() __attribute__((always_inline)) {
        epoch++;
        const int par = epoch & 1, cg = sh_cg;
        double2 v[J];
#pragma unroll
        for (int j = 0; j < J; j++) v[j] = x[0][j];
        // v = x[cg] as a chain of register selects.  The empty asm keeps hipcc from rewriting the chain
        // into a dynamically indexed load, which would move all my rows from registers to scratch.
#pragma unroll
        for (int g = 1; g < R; g++) {
#pragma unroll
            for (int j = 0; j < J; j++) {
                double ax = x[g][j].x, ay = x[g][j].y;
                asm volatile("" : "+v"(ax), "+v"(ay));
                v[j].x = g == cg ? ax : v[j].x;
                v[j].y = g == cg ? ay : v[j].y;
            }
        }
        double *dst = d.rc_rows[par] + (size_t)b * pitch;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c0 = 2 * (tid + j * T);
            if (c0 < pitch) st16_sc1(dst + c0, v[j]);
        }
        if (tid == cg) st_sc1(d.rc_key[par] + b, my_rhs); // the candidate row's RHS entry (lane cg)
        if (tid == 0) // the key travels next to the flag: one 16-byte record per workgroup
            __hip_atomic_store(d.rc_flag[par] + 2 * b, (unsigned long long)__double_as_longlong(sh_ck), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
        __syncthreads();                                  // ... before ONE lane raises the flag
        if (tid == 0)
            __hip_atomic_store(d.rc_flag[par] + 2 * b + 1, ((unsigned long long)epoch << 32) | (unsigned)sh_ci,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // entries of my rows in column la, as the rows are now -> sh_val[0..R)
    auto column_la = [&]() __attribute__((always_inline)) {
        const int ula = (la - 1) >> 1, ela = (la - 1) & 1;
        if (la > 0 && tid == ula % T) {
#pragma unroll
            for (int g = 0; g < R; g++)
#pragma unroll
                for (int j = 0; j < J; j++)
                    if (j == ula / T) sh_val[g] = elem(x[g][j], ela);
        }
        __syncthreads();
    };
    int done = 0, term = RUNNING;
    double term_result = NAN;
    bool stop = false;
    // loop bound, optimality: checked before every exchange (src/simplex.ts:69,109 and :80)
    auto check = [&]() __attribute__((always_inline)) {
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
    column_la();
    check();
    if (!stop) {
        candidate(phase);
        publish();
    }
    // (single back edge, single exit: every `stop` is a flag, so the rows stay in one set of registers)
    while (!stop) {
        // ---------------- gather everyone's candidate -------------------------------------------
        const int par = epoch & 1;
        KI c = {INFINITY, INT_MAX};
        if (tid < NB) {
            // The key word was stored and drained before the flag word of the same 16-byte record,
            // and is read AFTER the poll matched (program order of two sc1 loads of one lane).
            unsigned long long f = 0;
            unsigned spins = 0;
            for (;;) {
                f = __hip_atomic_load(d.rc_flag[par] + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(f >> 32) == epoch) break;
                if (++spins > (1u << 22) || __hip_atomic_load(d.rc_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    sh_fail = 1;
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            c.i = (int)(unsigned)f;
            c.k = __longlong_as_double((long long)__hip_atomic_load(d.rc_flag[par] + 2 * tid, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT));
        }
        c = block_argmin<T>(c, sk, si, slot); // (its barrier is the one the polling waves join)
        slot ^= 1;
        if (sh_fail) return; // uniform: written before the barrier above
        if (c.i == INT_MAX) {
            if (phase == 1) { // :120 phase 1 is over: same tableau, now the min-ratio exchange
                phase = 2;
                iter = 0.0;
                check();
                if (!stop) {
                    column_la(); // my rows are complete here: their entries of column la
                    candidate(2);
                    publish();
                }
            } else {
                term = YALPS_UNBOUNDED; // :96
                term_result = (double)la;
                stop = true;
            }
            continue;
        }
        const int row = c.i, owner = row % NB;
        // ---------------- the winner's raw row (sc1 loads only) ----------------------------------
        const double *src = d.rc_rows[par] + (size_t)owner * pitch;
        const double rhs_row = ld_sc1(d.rc_key[par] + owner);
        double2 pv[J];
        ld16_sc1<J>(pv, src, cofs);
        int col = la;
        if (phase == 1) { // :123-134
            KI e = {INFINITY, INT_MAX};
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const double coefficient = elem(pv[j], k);
                    if (c0 + k < n && coefficient < -precision) {
                        const double ratio = -elem(o[j], k) / coefficient;
                        if (ratio > -INFINITY && ki_better(-ratio, c0 + k + 1, e.k, e.i)) {
                            e.k = -ratio;
                            e.i = c0 + k + 1;
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
            col = e.i;
        }
        // ---------------- pivot (src/simplex.ts:5-39) on my registers ----------------------------
        // Order: everything the NEXT exchange needs first (objective replica -> la, my rows' entries
        // of column la and RHS -> my candidate, that one row), publish, and only then the other rows:
        // their elimination overlaps the time the flags take to travel.
        const int ucol = (col - 1) >> 1, ecol = (col - 1) & 1, col_tid = ucol % T, col_j = ucol / T;
        if (tid == col_tid) { // pivot-column entries of my rows, of the objective row, the quotient
#pragma unroll
            for (int j = 0; j < J; j++)
                if (j == col_j) {
#pragma unroll
                    for (int g = 0; g < R; g++) sh_val[g] = elem(x[g][j], ecol);
                    sh_val[R] = elem(o[j], ecol);
                    sh_val[R + 1] = elem(pv[j], ecol);
                }
        }
        __syncthreads();
        const double q = sh_val[R + 1], coef0 = sh_val[R];
        double cf[R]; // uniform: pivot-column entry of each of my rows
#pragma unroll
        for (int g = 0; g < R; g++) cf[g] = sh_val[g];
        unsigned nzmask = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double v = elem(pv[j], k);
                const bool nz = fabs(v) > 1e-16;
                pv[j] = with_elem(pv[j], k, nz ? v / q : 0.0);
                if (nz) nzmask |= 1u << (2 * j + k);
            }
        }
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const int lslot = owner == b ? row / NB : -1; // my register slot of the pivot row, if I own it
        // the R + 2 divisions of the pivot column (one per lane of wave 0, not R+2 per lane)
        if (tid < R + 2) sh_nq[tid] = tid == R + 1 ? 1.0 / q : -sh_val[tid] / q;
        if (my_live) { // RHS entry of my row (:33 at column 0)
            const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
            double my_coef = 0.0;
#pragma unroll
            for (int g = 0; g < R; g++)
                if (tid == g) my_coef = cf[g];
            if (tid == lslot)
                my_rhs = pn_rhs;
            else if (fabs(my_coef) > 1e-16 && nz_rhs) {
                const double prod = my_coef * pn_rhs;
                my_rhs = my_rhs - prod;
            }
        }
        const bool touched0 = fabs(coef0) > 1e-16;
        if (touched0) { // my replica of the objective row (branch-free over the lane's columns)
#pragma unroll
            for (int j = 0; j < J; j++) {
                const double px = coef0 * pv[j].x, py = coef0 * pv[j].y;
                const double nx = o[j].x - px, ny = o[j].y - py;
                o[j].x = (nzmask & (1u << (2 * j))) ? nx : o[j].x;
                o[j].y = (nzmask & (1u << (2 * j + 1))) ? ny : o[j].y;
            }
        }
        __syncthreads(); // sh_nq visible; sh_val (pivot column) consumed
        if (touched0 && tid == col_tid) {
#pragma unroll
            for (int j = 0; j < J; j++)
                if (j == col_j) o[j] = with_elem(o[j], ecol, sh_nq[R]);
        }
        // my rows, fully, as pivot() leaves them: slot `only` (only_it = true) or all slots but it
        auto finish_rows = [&](int only, bool only_it) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < R; g++) { // (g must stay a compile-time index: the rows are registers)
            if ((g == only) != only_it) continue;
            if (g == lslot) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    x[g][j] = pv[j];
                    if (tid == col_tid && j == col_j) x[g][j] = with_elem(x[g][j], ecol, sh_nq[R + 1]); // :25
                }
            } else if (b + NB * g < h && fabs(cf[g]) > 1e-16) { // :31 (uniform per row)
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const double px = cf[g] * pv[j].x, py = cf[g] * pv[j].y;
                    const double nx = x[g][j].x - px, ny = x[g][j].y - py;
                    x[g][j].x = (nzmask & (1u << (2 * j))) ? nx : x[g][j].x;
                    x[g][j].y = (nzmask & (1u << (2 * j + 1))) ? ny : x[g][j].y;
                    if (tid == col_tid && j == col_j) x[g][j] = with_elem(x[g][j], ecol, sh_nq[g]); // :36
                }
            }
            }
        };
        iter += 1.0;
        pivots += 1;
        done += 1;
        price(); // la of the next pivot, from the updated objective replica
        check();
        if (!stop) {
            if (phase == 2) {
                // my rows' entries of column la AFTER this pivot, computed by the lane that holds them
                const int ula = (la - 1) >> 1, ela = (la - 1) & 1;
                if (tid == ula % T) {
#pragma unroll
                    for (int j = 0; j < J; j++)
                        if (j == ula / T) {
                            const double p = elem(pv[j], ela);
                            const bool nz = (nzmask >> (2 * j + ela)) & 1u;
#pragma unroll
                            for (int g = 0; g < R; g++) {
                                double v = elem(x[g][j], ela);
                                if (g == lslot)
                                    v = la == col ? sh_nq[R + 1] : p;
                                else if (b + NB * g < h && fabs(cf[g]) > 1e-16) {
                                    if (la == col)
                                        v = sh_nq[g];
                                    else if (nz) {
                                        const double prod = cf[g] * p;
                                        v = v - prod;
                                    }
                                }
                                sh_val[g] = v;
                            }
                        }
                }
                __syncthreads();
            }
            candidate(phase);
            const int cg = sh_cg;
            finish_rows(cg, true);
            publish();
            finish_rows(cg, false);
        } else {
            finish_rows(-1, false);
        }
        if (b == 0 && tid == 0) { // basis bookkeeping, :7-12, in LDS (off the critical path)
            int *var = sh_perm, *pos = sh_perm + d.perm_len;
            const int leaving = var[w + row], entering = var[col];
            var[w + row] = entering;
            var[col] = leaving;
            pos[leaving] = col;
            pos[entering] = w + row;
        }
        // (no barrier here: the next write to sh_val / sh_nq comes after the gather's barrier, which every
        // wave reaches only after it has finished reading them)
    }

    // ---------------- leave: tableau to the other buffer, state, basis ---------------------------
    double *matB = d.mat[mbuf ^ 1];
    double *rhsB = d.rhs[mbuf ^ 1];
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int r = b + NB * g;
        if (r < h) {
            double *mr = matB + (size_t)r * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 < pitch) *reinterpret_cast<double2 *>(mr + c0) = x[g][j];
            }
        }
    }
    if (my_live) rhsB[my_r] = my_rhs;
    if (b == 0) {
        for (int i = tid; i < d.perm_len; i += T) {
            d.var[i] = sh_perm[i];
            d.pos[i] = sh_perm[d.perm_len + i];
        }
        if (tid == 0) {
            if (term == YALPS_OPTIMAL) term_result = round_to_precision(my_rhs, precision); // lane 0 = row 0
            Sout->status = term;
            Sout->phase = phase;
            Sout->bootstrap = 1; // the streaming kernel would have to re-scan
            Sout->la = 0;
            Sout->pbuf = 0;
            Sout->mbuf = mbuf ^ 1;
            Sout->pause = 0;
            Sout->dec_valid = 0;
            Sout->dec_row = 0;
            Sout->dec_col = 0;
            Sout->swap_valid = 0;
            Sout->swap_row = 0;
            Sout->swap_col = 0;
            Sout->pad_ = 0;
            Sout->hist_len = 0;
            Sout->iter = iter;
            Sout->result = term_result;
            Sout->pivots = pivots;
        }
    }
}

// ------------------------------------------------------------------------------------------
// batch_kernel: many independent branch-and-cut nodes at once, ONE workgroup per node
// (BASELINE config 4).  A node's LP is the root's optimal tableau plus one
// row per cut (src/branchAndCut.ts:22-61 `applyCuts`), re-solved with simplex() (:127).  The
// root stays resident in HBM; every workgroup builds its node's tableau in its own workspace and
// runs the whole two-phase loop there with workgroup-local synchronisation only -- nodes are
// independent, there is no cross-workgroup communication and no host round trip.
// Per pivot: the pivot column is gathered into LDS first (so rows can then be updated in place),
// the pivot row is normalised into LDS (FLUSHED marks entries pivot() zeroed), the sweep gives
// every lane fixed 16-byte column units and walks the rows.
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

template <int T>
__global__ __launch_bounds__(T) void batch_kernel(BatchDesc d) {
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    extern __shared__ double sh_dyn[]; // colbuf[hmax], prow[pitch]
    double *colbuf = sh_dyn, *prow = sh_dyn + d.hmax;

    const int tid = threadIdx.x, node = blockIdx.x;
    const int w = d.w, n = d.n, pitch = d.pitch, h0 = d.h0;
    const double precision = d.precision, max_pivots = d.max_pivots;
    double *mat = d.ws_mat + (size_t)node * d.hmax * pitch;
    double *rhs = d.ws_rhs + (size_t)node * d.hmax;
    int32_t *pos = d.ws_pos + (size_t)node * d.permmax;
    int32_t *var = d.ws_var + (size_t)node * d.permmax;
    const int c_lo = d.cut_off[node], ncuts = d.cut_off[node + 1] - c_lo;
    const int h = h0 + ncuts;
    const int units = pitch / 2;
    int slot = 0;

    // ---- applyCuts (src/branchAndCut.ts:22-61) ----
    for (int r = 0; r < h0; r++) {
        const double *src = d.root_mat + (size_t)r * pitch;
        double *dst = mat + (size_t)r * pitch;
        for (int u = tid; u < units; u += T)
            *reinterpret_cast<double2 *>(dst + 2 * u) = *reinterpret_cast<const double2 *>(src + 2 * u);
    }
    for (int r = tid; r < h0; r += T) rhs[r] = d.root_rhs[r];
    for (int i = 0; i < ncuts; i++) {
        const double sign = (double)d.cut_sign[c_lo + i], value = d.cut_val[c_lo + i];
        const int p = d.root_pos[d.cut_var[c_lo + i]];
        double *dst = mat + (size_t)(h0 + i) * pitch;
        if (p < w) { // non-basic at the root: sign * x <= sign * value   (:32-35)
            for (int c = tid; c < pitch; c += T) dst[c] = (c == p - 1) ? sign : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * value;
        } else { // basic in root row p - w: substitute that row   (:36-42)
            const double *src = d.root_mat + (size_t)(p - w) * pitch;
            for (int c = tid; c < pitch; c += T) dst[c] = c < n ? -sign * src[c] : 0.0;
            if (tid == 0) rhs[h0 + i] = sign * (value - d.root_rhs[p - w]);
        }
    }
    for (int i = tid; i < w + h; i += T) { // :46-52
        pos[i] = i < w + h0 ? d.root_pos[i] : i;
        var[i] = i < w + h0 ? d.root_var[i] : i;
    }
    __syncthreads();

    // ---- simplex(): src/simplex.ts:106-142 then :66-103 ----
    int phase = 1, status = YALPS_CYCLED;
    double iter = 0.0, result = NAN;
    long long pivots = 0;
    for (;;) {
        if (!(iter < max_pivots)) break; // "cycled"
        int row = 0, col = 0;
        if (phase == 1) {
            KI c = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += T) {
                const double v = rhs[r];
                if (v < -precision && ki_better(v, r, c.k, c.i)) {
                    c.k = v;
                    c.i = r;
                }
            }
            c = block_argmin<T>(c, sk, si, slot);
            slot ^= 1;
            if (c.i == INT_MAX) {
                phase = 2;
                iter = 0.0;
                continue;
            }
            row = c.i;
            const double *mrow = mat + (size_t)row * pitch;
            KI e = {INFINITY, INT_MAX};
            for (int cc = tid; cc < n; cc += T) {
                const double coefficient = mrow[cc];
                if (coefficient < -precision) {
                    const double ratio = -mat[cc] / coefficient;
                    if (ratio > -INFINITY && ki_better(-ratio, cc + 1, e.k, e.i)) {
                        e.k = -ratio;
                        e.i = cc + 1;
                    }
                }
            }
            e = block_argmin<T>(e, sk, si, slot);
            slot ^= 1;
            if (e.i == INT_MAX) {
                status = YALPS_INFEASIBLE;
                break;
            }
            col = e.i;
        } else {
            KI pr = {INFINITY, INT_MAX};
            for (int cc = tid; cc < n; cc += T) {
                const double rc = mat[cc];
                if (rc > precision && ki_better(-rc, cc + 1, pr.k, pr.i)) {
                    pr.k = -rc;
                    pr.i = cc + 1;
                }
            }
            pr = block_argmin<T>(pr, sk, si, slot);
            slot ^= 1;
            if (pr.i == INT_MAX) {
                status = YALPS_OPTIMAL;
                result = round_to_precision(rhs[0], precision);
                break;
            }
            col = pr.i;
            KI c = {INFINITY, INT_MAX};
            for (int r = 1 + tid; r < h; r += T) {
                const double value = mat[(size_t)r * pitch + col - 1];
                if (value <= precision) continue;
                const double ratio = rhs[r] / value;
                if (!(ratio < INFINITY)) continue;
                const double key = (ratio <= precision) ? -INFINITY : ratio;
                if (ki_better(key, r, c.k, c.i)) {
                    c.k = key;
                    c.i = r;
                }
            }
            c = block_argmin<T>(c, sk, si, slot);
            slot ^= 1;
            if (c.i == INT_MAX) {
                status = YALPS_UNBOUNDED;
                result = (double)col;
                break;
            }
            row = c.i;
        }
        // ---- pivot(row, col): src/simplex.ts:5-39 ----
        for (int r = tid; r < h; r += T) colbuf[r] = mat[(size_t)r * pitch + col - 1];
        __syncthreads();
        const double q = colbuf[row], rhs_row = rhs[row];
        double *mrow = mat + (size_t)row * pitch;
        for (int c = tid; c < pitch; c += T) {
            const double v = mrow[c];
            const bool nz = fabs(v) > 1e-16;
            const double pn = nz ? v / q : 0.0;
            mrow[c] = (c == col - 1) ? 1.0 / q : pn;
            prow[c] = nz ? pn : __longlong_as_double((long long)FLUSHED);
        }
        __syncthreads(); // (also: everybody has read rhs[row] before it changes)
        const bool nz_rhs = fabs(rhs_row) > 1e-16;
        const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
        for (int r = tid; r < h; r += T) {
            if (r == row) {
                rhs[r] = pn_rhs;
            } else if (nz_rhs && fabs(colbuf[r]) > 1e-16) {
                const double prod = colbuf[r] * pn_rhs;
                rhs[r] = rhs[r] - prod;
            }
        }
        for (int u = tid; u < units; u += T) {
            const double2 p = *reinterpret_cast<const double2 *>(prow + 2 * u);
            const bool f0 = (unsigned long long)__double_as_longlong(p.x) != FLUSHED;
            const bool f1 = (unsigned long long)__double_as_longlong(p.y) != FLUSHED;
            const bool has_col = (col - 1) >> 1 == u;
#pragma unroll 4
            for (int r = 0; r < h; r++) {
                const double coef = colbuf[r];
                if (r == row || !(fabs(coef) > 1e-16)) continue; // uniform
                double2 *xp = reinterpret_cast<double2 *>(mat + (size_t)r * pitch + 2 * u);
                double2 x = *xp;
                if (f0) {
                    const double prod = coef * p.x;
                    x.x = x.x - prod;
                }
                if (f1) {
                    const double prod = coef * p.y;
                    x.y = x.y - prod;
                }
                if (has_col) {
                    const double nq = -coef / q;
                    if ((col - 1) & 1)
                        x.y = nq;
                    else
                        x.x = nq;
                }
                *xp = x;
            }
        }
        if (tid == 0) { // :7-12
            const int leaving = var[w + row], entering = var[col];
            var[w + row] = entering;
            var[col] = leaving;
            pos[leaving] = col;
            pos[entering] = w + row;
        }
        iter += 1.0;
        pivots += 1;
        __syncthreads();
    }
    if (tid == 0) {
        d.status[node] = status;
        d.result[node] = result;
        d.pivots[node] = pivots;
        d.height[node] = h;
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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
using KernelFn = void (*)(Desc, int, int, int, const double *);

struct Variant {
    int T, J, R;
    KernelFn fn;
};

#define VARIANT(T, J, R, D) {T, J, R, pivot_kernel<T, J, R, D>}
const Variant kVariants[] = {
    VARIANT(256, 1, 4, 4),   VARIANT(256, 1, 9, 9),  VARIANT(256, 1, 16, 8),  VARIANT(256, 2, 4, 4),
    VARIANT(256, 2, 8, 8),   VARIANT(1024, 1, 4, 4), VARIANT(1024, 1, 9, 9), VARIANT(1024, 1, 16, 8),
    VARIANT(1024, 2, 4, 4),  VARIANT(1024, 2, 8, 8), VARIANT(1024, 4, 4, 2), VARIANT(1024, 8, 2, 1),
};
#undef VARIANT

struct WVariant {
    int T, J;
    KernelFn fn;
};
const WVariant kWide[] = {
    {256, 1, wide_kernel<256, 1>},   {256, 2, wide_kernel<256, 2>},   {1024, 1, wide_kernel<1024, 1>},
    {1024, 2, wide_kernel<1024, 2>}, {1024, 4, wide_kernel<1024, 4>}, {1024, 8, wide_kernel<1024, 8>},
};

using ResidentFn = void (*)(Desc, int, int);
struct RVariant {
    int T, J, R;
    ResidentFn fn;
};
#define RVARIANT(T, J, R) {T, J, R, resident_kernel<T, J, R>}
const RVariant kResident[] = {
    // few, fat lanes: the loop is latency-bound, and <= 8 waves per CU leave each lane 256 VGPRs
    RVARIANT(256, 1, 4), RVARIANT(256, 1, 9), RVARIANT(256, 1, 16), RVARIANT(256, 2, 4), RVARIANT(256, 2, 9),
    RVARIANT(256, 2, 16), RVARIANT(512, 2, 4), RVARIANT(512, 2, 9), RVARIANT(512, 2, 16), RVARIANT(512, 4, 4),
    RVARIANT(512, 4, 9),
};
#undef RVARIANT
constexpr int RESIDENT_CHUNK = 4096; // pivots per launch of the resident kernel (bounds its run time)

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

int env_int(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return (e && *e) ? std::atoi(e) : dflt;
}

} // namespace

struct yalps_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool eager = false;
    bool nt_stores = false;
    bool resident = true; // use the on-chip resident kernel when the tableau fits (YALPS_HIP_RESIDENT=0: never)
    int resident_chunk = RESIDENT_CHUNK; // pivots per resident launch (YALPS_HIP_RESIDENT_CHUNK)
    int resident_fault = 0; // test hook: treat the N-th resident launch as failed (YALPS_HIP_RESIDENT_FAULT=N)
    int num_cus = 256;
    int max_blocks = 256; // workgroups per launch (one per CU by default)
};

struct yalps_tableau {
    yalps_ctx *ctx = nullptr;
    Desc d{};
    int32_t height = 0;
    int cur = 0; // tableau buffer holding the current tableau
    int shard_parity = 0;
    RVariant rvar{0, 0, 0, nullptr}; // resident kernel variant, fn == nullptr: tableau does not fit
    int last_path = 0;               // what the last solve ran: 1 resident, 2 streaming, 3 both
    int64_t last_launches = 0;       // kernel launches of the last solve that did work (resident: chunks)
    size_t rshmem = 0;
    int32_t *perm_backup = nullptr; // basis before the resident launch in flight (restored if it fails)
    int32_t perm_backup_len = 0;
    int32_t perm_len = 0; // entries of pos / var (width + GLOBAL height)
    Variant var{};
    KernelFn wfn = nullptr; // wide_kernel variant used for FUSED / APPLY / SHARD launches when the tableau is
                            // too wide or too tall for pivot_kernel's register-resident batches
    size_t wshmem = 0;
    int nb = 1;
    hipGraph_t graph[2] = {nullptr, nullptr}; // [0] fused, [1] decide/apply (checkCycles)
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    YState *host_state = nullptr; // pinned, 4 rotating slots
    hipEvent_t slot_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *hist[2] = {nullptr, nullptr};
    int64_t hist_cap = 0;
};

namespace {

void launch_one(yalps_tableau *t, int parity, int mode, int force, const double *gather = nullptr) {
    const int grid = mode == MODE_DECIDE ? 1 : t->nb;
    if (t->wfn && mode != MODE_DECIDE)
        t->wfn<<<dim3(grid), dim3(t->var.T), t->wshmem, t->ctx->stream>>>(t->d, parity, mode, force, gather);
    else
        t->var.fn<<<dim3(grid), dim3(t->var.T), 0, t->ctx->stream>>>(t->d, parity, mode, force, gather);
}

int launch_batch(yalps_tableau *t, int which) {
    for (int i = 0; i < LAUNCHES_PER_GRAPH; i++) {
        const int mode = which == 0 ? MODE_FUSED : ((i & 1) ? MODE_DECIDE : MODE_APPLY);
        launch_one(t, i & 1, mode, t->ctx->nt_stores ? 64 : 0);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ensure_graph(yalps_tableau *t, int which) {
    if (t->graph_exec[which] || t->ctx->eager) return 0;
    hipStream_t s = t->ctx->stream;
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = launch_batch(t, which);
    hipError_t e = hipStreamEndCapture(s, &t->graph[which]);
    if (rc) return rc;
    HIP_TRY(e);
    HIP_TRY(hipGraphInstantiate(&t->graph_exec[which], t->graph[which], nullptr, nullptr, 0));
    return 0;
}

int run_batch(yalps_tableau *t, int which) {
    if (t->ctx->eager) return launch_batch(t, which);
    HIP_TRY(hipGraphLaunch(t->graph_exec[which], t->ctx->stream));
    return 0;
}

int grow_history(yalps_tableau *t, int64_t need, int64_t keep) {
    int64_t cap = t->hist_cap ? t->hist_cap : 4096;
    while (cap < need) cap *= 2;
    if (cap == t->hist_cap) return 0;
    hipStream_t s = t->ctx->stream;
    for (int k = 0; k < 2; k++) {
        int32_t *nb = nullptr;
        HIP_TRY(hipMalloc(&nb, sizeof(int32_t) * (size_t)cap));
        if (t->hist[k] && keep > 0)
            HIP_TRY(hipMemcpyAsync(nb, t->hist[k], sizeof(int32_t) * (size_t)keep, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (t->hist[k]) HIP_TRY(hipFree(t->hist[k]));
        t->hist[k] = nb;
    }
    t->hist_cap = cap;
    return 0;
}

int init_state(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles) {
    hipStream_t s = t->ctx->stream;
    if (checkCycles && !t->hist_cap) {
        int rc = grow_history(t, 4096, 0);
        if (rc) return rc;
    }
    if (t->cur != 0) { // FUSED graphs derive the buffer index from the launch parity: start from 0
        const Desc &d = t->d;
        HIP_TRY(hipMemcpyAsync(d.mat[0], d.mat[1], sizeof(double) * (size_t)d.pitch * t->height,
                               hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(d.rhs[0], d.rhs[1], sizeof(double) * (size_t)t->height, hipMemcpyDeviceToDevice, s));
        t->cur = 0;
    }
    YConst hc;
    std::memset(&hc, 0, sizeof hc);
    hc.height = t->height;
    hc.check_cycles = checkCycles ? 1 : 0;
    hc.hist_cap = t->hist_cap;
    hc.hist_leaving = t->hist[0];
    hc.hist_entering = t->hist[1];
    hc.precision = precision;
    hc.max_pivots = maxPivots;
    HIP_TRY(hipMemcpyAsync(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice, s));
    YState *hs = &t->host_state[0];
    std::memset(hs, 0, sizeof(YState));
    hs->status = RUNNING;
    hs->phase = 1;
    hs->bootstrap = 1;
    hs->mbuf = t->cur;
    hs->result = NAN;
    HIP_TRY(hipMemcpyAsync(t->d.st, hs, sizeof(YState), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // slot 0 is reused by the polling loop
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

static int32_t ctx_create(int32_t device, void *ext_stream, bool adopt, yalps_ctx **out) {
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
    *out = c;
    if (adopt) {
        c->stream = static_cast<hipStream_t>(ext_stream); // e.g. torch's current stream (may be the null stream)
        c->own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    c->eager = env_int("YALPS_HIP_EAGER", 0) != 0;
    c->nt_stores = env_int("YALPS_HIP_NT", 1) != 0;
    c->resident = env_int("YALPS_HIP_RESIDENT", 1) != 0;
    c->resident_chunk = env_int("YALPS_HIP_RESIDENT_CHUNK", RESIDENT_CHUNK);
    if (c->resident_chunk < 1) c->resident_chunk = 1;
    c->resident_fault = env_int("YALPS_HIP_RESIDENT_FAULT", 0);
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; // measured 1-2 % faster in the pivot loop at 2049^2 and 4097^2
    c->max_blocks = env_int("YALPS_HIP_BLOCKS", prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    if (c->max_blocks < 1) c->max_blocks = 1;
    if (c->max_blocks > MAX_BLOCKS) c->max_blocks = MAX_BLOCKS;
    *out = c;
    return 0;
}

static int32_t ctx_create_guarded(int32_t device, void *stream, bool adopt, yalps_ctx **out) {
    if (!out) return fail(YALPS_E_ARG, "yalps_ctx_create: out is NULL");
    *out = nullptr;
    const int32_t rc = ctx_create(device, stream, adopt, out);
    if (rc && *out) {
        const std::string why = g_err;
        yalps_ctx_destroy(*out);
        *out = nullptr;
        g_err = why;
    }
    return rc;
}

int32_t yalps_ctx_create(int32_t device, yalps_ctx **out) { return ctx_create_guarded(device, nullptr, false, out); }

int32_t yalps_ctx_create_on_stream(int32_t device, void *hip_stream, yalps_ctx **out) {
    return ctx_create_guarded(device, hip_stream, true, out);
}

void yalps_ctx_destroy(yalps_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int32_t tableau_create_impl(yalps_ctx *ctx, int32_t width, int32_t hcap, yalps_tableau **out) {
    if ((int64_t)width + hcap > INT32_MAX / 2) return fail(YALPS_E_ARG, "tableau too large");
    const int n = width - 1, units = (n + 1) / 2;
    // kernel variant: lanes x units-per-lane must span the row; rows in flight sized so that
    // one workgroup per CU covers the tableau in one batch when it can
    int T = units <= 512 ? 256 : 1024;
    int J = 1;
    while (T * J < units) J *= 2;
    if (J > 8) return fail(YALPS_E_ARG, "width > 16385 columns is not supported by this build");
    HIP_TRY(hipSetDevice(ctx->device));
    yalps_tableau *t = new yalps_tableau();
    t->ctx = ctx;
    *out = t; // reachable from now on: the wrapper frees a half-built object on failure
    const int forceR = env_int("YALPS_HIP_ROWS", 0);
    // spread the rows over all workgroups; rows in flight per lane R >= rows per workgroup if any
    // variant allows it (one batch per launch), else the largest R (several batches)
    t->nb = hcap < ctx->max_blocks ? hcap : ctx->max_blocks;
    const int rows_per_block = (hcap + t->nb - 1) / t->nb;
    const Variant *pick = nullptr;
    for (const Variant &v : kVariants) {
        if (v.T != T || v.J != J) continue;
        pick = &v; // candidates are listed by increasing R
        if (v.R >= (forceR ? forceR : rows_per_block)) break;
    }
    t->var = *pick;
    if (J >= 4 || rows_per_block > pick->R || env_int("YALPS_HIP_WIDE", 0)) {
        for (const WVariant &v : kWide)
            if (v.T == T && v.J == J) t->wfn = v.fn;
        t->wshmem = sizeof(double) * ((size_t)((n + 15) / 16 * 16 < 16 ? 16 : (n + 15) / 16 * 16) + 2 * (size_t)rows_per_block);
        if (t->wshmem > 150 * 1024) return fail(YALPS_E_ARG, "tableau too wide for this build");
        if (t->wshmem > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(t->wfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)t->wshmem));
    }

    Desc &d = t->d;
    d.w = width;
    d.n = n;
    d.hcap = hcap;
    d.nb = t->nb;
    d.nshards = 1;
    d.shard_rank = 0;
    d.row_base = 0;
    for (int k = 0; k <= MAX_SHARDS; k++) d.bounds[k] = k == 0 ? 0 : INT_MAX;
    t->perm_len = width + hcap;
    d.pitch = (n + 15) / 16 * 16; // 128-byte rows
    if (d.pitch < 16) d.pitch = 16;
    const size_t mat_bytes = sizeof(double) * (size_t)d.pitch * hcap;
    hipStream_t s = ctx->stream;
    for (int k = 0; k < 2; k++) {
        HIP_TRY(hipMalloc(&d.mat[k], mat_bytes));
        HIP_TRY(hipMemsetAsync(d.mat[k], 0, mat_bytes, s));
        HIP_TRY(hipMalloc(&d.rhs[k], sizeof(double) * (size_t)hcap));
    }
    HIP_TRY(hipMalloc(&d.pos, sizeof(int32_t) * (size_t)(width + hcap)));
    HIP_TRY(hipMalloc(&d.var, sizeof(int32_t) * (size_t)(width + hcap)));
    HIP_TRY(hipMalloc(&d.st, sizeof(YState) * 2));
    HIP_TRY(hipMemsetAsync(d.st, 0, sizeof(YState) * 2, s));
    HIP_TRY(hipMalloc(&d.cst, sizeof(YConst)));
    for (int k = 0; k < 2; k++) {
        HIP_TRY(hipMalloc(&d.part_ratio[k], sizeof(Part) * MAX_BLOCKS));
        HIP_TRY(hipMalloc(&d.part_rhs[k], sizeof(Part) * MAX_BLOCKS));
    }
    // resident (on-chip) solver: needs every workgroup co-resident (one per CU) and the rows of a
    // workgroup in registers
    d.perm_len = t->perm_len;
    if (t->nb <= ctx->num_cus) {
        // (16 waves per CU were tried for 2049^2: <1024,1,9> spills at the 128-VGPR cap and its barriers
        // cost more: 92 K pivots/s against 144 K for <512,2,9>)
        const int rT = units <= 512 ? 256 : 512;
        const int rJ = units <= 256 ? 1 : units <= 1024 ? 2 : 4;
        for (const RVariant &v : kResident) {
            if (units > 2048 || v.T != rT || v.J != rJ || v.R < rows_per_block) continue;
            t->rvar = v;
            break;
        }
    }
    if (t->rvar.fn) {
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipMalloc(&d.rc_rows[k], sizeof(double) * (size_t)t->nb * d.pitch));
            HIP_TRY(hipMalloc(&d.rc_key[k], sizeof(double) * ((size_t)t->nb + 8)));
            HIP_TRY(hipMalloc(&d.rc_flag[k], sizeof(unsigned long long) * 2 * (size_t)t->nb));
        }
        HIP_TRY(hipMalloc(&d.rc_err, sizeof(int32_t)));
    }
    HIP_TRY(hipHostMalloc(&t->host_state, sizeof(YState) * 4, hipHostMallocDefault));
    for (int k = 0; k < 4; k++) HIP_TRY(hipEventCreateWithFlags(&t->slot_ev[k], hipEventDisableTiming));
    HIP_TRY(hipStreamSynchronize(s));
    *out = t;
    return 0;
}

int32_t yalps_tableau_create(yalps_ctx *ctx, int32_t width, int32_t hcap, yalps_tableau **out) {
    if (!ctx || !out || width < 1 || hcap < 1) return fail(YALPS_E_ARG, "yalps_tableau_create: bad argument");
    *out = nullptr;
    const int32_t rc = tableau_create_impl(ctx, width, hcap, out);
    if (rc && *out) {
        const std::string why = g_err;
        yalps_tableau_destroy(*out);
        *out = nullptr;
        g_err = why;
    }
    return rc;
}

void yalps_tableau_destroy(yalps_tableau *t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (int k = 0; k < 2; k++) {
        if (t->graph_exec[k]) (void)hipGraphExecDestroy(t->graph_exec[k]);
        if (t->graph[k]) (void)hipGraphDestroy(t->graph[k]);
    }
    Desc &d = t->d;
    void *bufs[] = {d.mat[0], d.mat[1], d.rhs[0], d.rhs[1], d.pos, d.var, d.st, d.cst, d.rc_rows[0], d.rc_rows[1], t->perm_backup,
                    d.rc_key[0], d.rc_key[1], d.rc_flag[0], d.rc_flag[1], d.rc_err, d.part_ratio[0], d.part_ratio[1], d.part_rhs[0], d.part_rhs[1],
                    t->hist[0], t->hist[1]};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (t->host_state) (void)hipHostFree(t->host_state);
    for (auto &e : t->slot_ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

int32_t yalps_tableau_height(const yalps_tableau *t) { return t ? t->height : 0; }

int32_t yalps_tableau_info(const yalps_tableau *t, char *buf, int32_t len) {
    if (!t || !buf || len < 1) return fail(YALPS_E_ARG, "yalps_tableau_info: bad argument");
    char res[96] = "none";
    if (t->rvar.fn)
        std::snprintf(res, sizeof res, "resident_kernel<%d,%d,%d> chunk=%d", t->rvar.T, t->rvar.J, t->rvar.R, RESIDENT_CHUNK);
    std::snprintf(buf, (size_t)len,
                  "streaming=pivot_kernel<%d,%d,%d> workgroups=%d resident=%s last_path=%s last_resident_launches=%lld",
                  t->var.T, t->var.J, t->var.R, t->nb, res,
                  t->last_path == 1 ? "resident" : t->last_path == 2 ? "streaming" : t->last_path == 3 ? "resident+streaming" : "none",
                  (long long)(t->last_path & 1 ? t->last_launches : 0));
    return 0;
}

int32_t yalps_tableau_upload(yalps_tableau *t, const double *matrix, int32_t height, const int32_t *pos,
                             const int32_t *var) {
    if (!t || !matrix || !pos || !var) return fail(YALPS_E_ARG, "yalps_tableau_upload: NULL argument");
    if (height < 1 || height > t->d.hcap) return fail(YALPS_E_ARG, "yalps_tableau_upload: height exceeds capacity");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    const Desc &d = t->d;
    // split: column 0 -> rhs[], columns 1..w-1 -> mat[][pitch]
    t->cur = 0;
    HIP_TRY(hipMemcpy2DAsync(d.rhs[0], sizeof(double), matrix, sizeof(double) * d.w, sizeof(double), height,
                             hipMemcpyHostToDevice, s));
    if (d.n > 0)
        HIP_TRY(hipMemcpy2DAsync(d.mat[0], sizeof(double) * d.pitch, matrix + 1, sizeof(double) * d.w,
                                 sizeof(double) * d.n, height, hipMemcpyHostToDevice, s));
    if (t->d.nshards > 1) return fail(YALPS_E_ARG, "yalps_tableau_upload: tableau is sharded; create a new one");
    const size_t nperm = sizeof(int32_t) * (size_t)(d.w + height);
    t->perm_len = d.w + height;
    t->d.perm_len = t->perm_len;
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
    if (matrix) {
        HIP_TRY(hipMemcpy2DAsync(matrix, sizeof(double) * d.w, d.rhs[t->cur], sizeof(double), sizeof(double),
                                 t->height, hipMemcpyDeviceToHost, s));
        if (d.n > 0)
            HIP_TRY(hipMemcpy2DAsync(matrix + 1, sizeof(double) * d.w, d.mat[t->cur], sizeof(double) * d.pitch,
                                     sizeof(double) * d.n, t->height, hipMemcpyDeviceToHost, s));
    }
    const size_t nperm = sizeof(int32_t) * (size_t)t->perm_len;
    if (pos) HIP_TRY(hipMemcpyAsync(pos, d.pos, nperm, hipMemcpyDeviceToHost, s));
    if (var) HIP_TRY(hipMemcpyAsync(var, d.var, nperm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_tableau_download_rhs(yalps_tableau *t, double *col0) {
    if (!t || !col0) return fail(YALPS_E_ARG, "yalps_tableau_download_rhs: NULL argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    HIP_TRY(hipMemcpyAsync(col0, t->d.rhs[t->cur], sizeof(double) * (size_t)t->height, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_tableau_copy(yalps_tableau *dst, const yalps_tableau *src) {
    if (!dst || !src) return fail(YALPS_E_ARG, "yalps_tableau_copy: NULL argument");
    if (dst->d.w != src->d.w || dst->d.hcap < src->height || dst->ctx != src->ctx || src->d.nshards > 1 ||
        dst->d.nshards > 1)
        return fail(YALPS_E_ARG, "yalps_tableau_copy: incompatible tableaux");
    HIP_TRY(hipSetDevice(dst->ctx->device));
    hipStream_t s = dst->ctx->stream;
    dst->cur = 0;
    HIP_TRY(hipMemcpyAsync(dst->d.mat[0], src->d.mat[src->cur], sizeof(double) * (size_t)src->d.pitch * src->height,
                           hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst->d.rhs[0], src->d.rhs[src->cur], sizeof(double) * (size_t)src->height,
                           hipMemcpyDeviceToDevice, s));
    const size_t nperm = sizeof(int32_t) * (size_t)(src->d.w + src->height);
    HIP_TRY(hipMemcpyAsync(dst->d.pos, src->d.pos, nperm, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst->d.var, src->d.var, nperm, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    dst->height = src->height;
    dst->perm_len = src->perm_len;
    dst->d.perm_len = src->perm_len;
    return 0;
}

int32_t yalps_tableau_solve(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles,
                            double *result_out, int64_t *pivots_out, float *gpu_ms_out) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_tableau_solve: no tableau uploaded");
    yalps_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int which = checkCycles ? 1 : 0;
    int rc = init_state(t, precision, maxPivots, checkCycles);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev0, s));
    YState fin;
    std::memset(&fin, 0, sizeof fin);
    bool finished = false;
    t->last_path = 0;
    t->last_launches = 0;

    // (a) the tableau fits on chip: persistent register-resident kernel, RESIDENT_CHUNK pivots per launch
    if (!checkCycles && c->resident && t->rvar.fn && t->d.nshards == 1) {
        const size_t shmem = sizeof(int32_t) * 2 * (size_t)t->perm_len;
        if (shmem != t->rshmem) {
            if (shmem > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(t->rvar.fn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            t->rshmem = shmem;
        }
        int parity = 0;
        int32_t *herr = reinterpret_cast<int32_t *>(&t->host_state[3]); // pinned scratch
        for (;;) {
            for (int k = 0; k < 2; k++)
                HIP_TRY(hipMemsetAsync(t->d.rc_flag[k], 0, sizeof(unsigned long long) * 2 * (size_t)t->nb, s));
            HIP_TRY(hipMemsetAsync(t->d.rc_err, 0, sizeof(int32_t), s));
            // the kernel rewrites the basis in place on exit: keep the old one until the launch is known good
            if (t->perm_backup_len < 2 * t->perm_len) {
                if (t->perm_backup) HIP_TRY(hipFree(t->perm_backup));
                HIP_TRY(hipMalloc(&t->perm_backup, sizeof(int32_t) * 2 * (size_t)t->perm_len));
                t->perm_backup_len = 2 * t->perm_len;
            }
            HIP_TRY(hipMemcpyAsync(t->perm_backup, t->d.var, sizeof(int32_t) * (size_t)t->perm_len, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(t->perm_backup + t->perm_len, t->d.pos, sizeof(int32_t) * (size_t)t->perm_len,
                                   hipMemcpyDeviceToDevice, s));
            t->rvar.fn<<<dim3(t->nb), dim3(t->rvar.T), shmem, s>>>(t->d, parity, c->resident_chunk);
            t->last_path |= 1;
            t->last_launches++;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(herr, t->d.rc_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(&t->host_state[1], t->d.st + (parity ^ 1), sizeof(YState), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (*herr || (c->resident_fault > 0 && t->last_launches == c->resident_fault)) {
                // a workgroup gave up waiting (grid not co-resident?): never again on this context;
                // carry on with the streaming kernel from the last consistent state (st[parity])
                c->resident = false;
                HIP_TRY(hipMemcpyAsync(t->d.var, t->perm_backup, sizeof(int32_t) * (size_t)t->perm_len, hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemcpyAsync(t->d.pos, t->perm_backup + t->perm_len, sizeof(int32_t) * (size_t)t->perm_len,
                                       hipMemcpyDeviceToDevice, s));
                YState last;
                HIP_TRY(hipMemcpy(&last, t->d.st + parity, sizeof(YState), hipMemcpyDeviceToHost));
                t->cur = last.mbuf;
                if (t->cur != 0) {
                    const Desc &d = t->d;
                    HIP_TRY(hipMemcpyAsync(d.mat[0], d.mat[1], sizeof(double) * (size_t)d.pitch * t->height,
                                           hipMemcpyDeviceToDevice, s));
                    HIP_TRY(hipMemcpyAsync(d.rhs[0], d.rhs[1], sizeof(double) * (size_t)t->height,
                                           hipMemcpyDeviceToDevice, s));
                    t->cur = 0;
                }
                last.mbuf = 0;
                last.bootstrap = 1;
                last.la = 0;
                last.pbuf = 0;
                HIP_TRY(hipMemcpy(t->d.st, &last, sizeof(YState), hipMemcpyHostToDevice));
                break;
            }
            if (t->host_state[1].status != RUNNING) {
                fin = t->host_state[1];
                finished = true;
                break;
            }
            parity ^= 1;
        }
    }

    // (b) general path: streaming kernel, one launch per pivot
    if (!finished) {
        rc = ensure_graph(t, which);
        if (rc) return rc;
        t->last_path |= 2;
    }
    // keep one batch in flight while the previous batch's state is inspected
    int issued = 0, checked = 0;
    while (!finished) {
        rc = run_batch(t, which);
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
            rc = grow_history(t, t->hist_cap * 2, now.hist_len);
            if (rc) return rc;
            YConst hc;
            HIP_TRY(hipMemcpy(&hc, t->d.cst, sizeof(YConst), hipMemcpyDeviceToHost));
            hc.hist_cap = t->hist_cap;
            hc.hist_leaving = t->hist[0];
            hc.hist_entering = t->hist[1];
            HIP_TRY(hipMemcpy(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice));
            now.pause = 0;
            HIP_TRY(hipMemcpy(t->d.st, &now, sizeof(YState), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (gpu_ms_out) HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    t->cur = fin.mbuf;
    if (result_out) *result_out = fin.result;
    if (pivots_out) *pivots_out = fin.pivots;
    return fin.status;
}

static int32_t set_decision(yalps_tableau *t, int32_t row, int32_t col) {
    int rc = init_state(t, 1e-8, INFINITY, 0);
    if (rc) return rc;
    YState *hs = &t->host_state[0];
    hs->bootstrap = 0;
    hs->phase = 1; // no pricing consequences: APPLY only normalises and eliminates
    hs->dec_valid = 1;
    hs->dec_row = row;
    hs->dec_col = col;
    HIP_TRY(hipMemcpy(t->d.st, hs, sizeof(YState), hipMemcpyHostToDevice));
    return 0;
}

int32_t yalps_tableau_pivot(yalps_tableau *t, int32_t row, int32_t col) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_tableau_pivot: no tableau uploaded");
    if (row < 0 || row >= t->height || col < 1 || col >= t->d.w) return fail(YALPS_E_ARG, "pivot out of range");
    HIP_TRY(hipSetDevice(t->ctx->device));
    int rc = set_decision(t, row, col);
    if (rc) return rc;
    launch_one(t, 0, MODE_APPLY, 0);
    flush_swap_kernel<<<dim3(1), dim3(64), 0, t->ctx->stream>>>(t->d, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    t->cur ^= 1; // the pivot wrote the other buffer
    return 0;
}

int32_t yalps_tableau_bench_sweep(yalps_tableau *t, int32_t row, int32_t col, int32_t launches, float *avg_us_out) {
    if (!t || t->height < 1 || launches < 1) return fail(YALPS_E_ARG, "yalps_tableau_bench_sweep: bad argument");
    if (row < 0 || row >= t->height || col < 1 || col >= t->d.w) return fail(YALPS_E_ARG, "pivot out of range");
    yalps_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = set_decision(t, row, col);
    if (rc) return rc;
    hipStream_t s = c->stream;
    // force=1: the state is left untouched, so every launch re-applies the same pivot
    const int fl = 1 | (c->nt_stores ? 64 : 0);
    for (int i = 0; i < 3; i++) launch_one(t, 0, MODE_APPLY, fl);
    HIP_TRY(hipEventRecord(c->ev0, s));
    for (int i = 0; i < launches; i++) launch_one(t, 0, MODE_APPLY, fl);
    HIP_TRY(hipEventRecord(c->ev1, s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (avg_us_out) *avg_us_out = ms * 1000.f / (float)launches;
    return 0;
}

// ---- row-sharded solve across GPUs (one process per GPU) ------------------------
// The library provides the per-rank steps, enqueued on the context's stream; the caller owns the
// collective between them (an all-gather of yalps_shard_slot_doubles() doubles per rank -- RCCL
// through torch.distributed in yalps_amd/sharded.py).
int32_t yalps_tableau_set_shard(yalps_tableau *t, int32_t rank, int32_t nranks, const int32_t *bounds,
                                int32_t global_height, const int32_t *pos, const int32_t *var) {
    if (!t || !bounds || !pos || !var || nranks < 1 || nranks > MAX_SHARDS || rank < 0 || rank >= nranks)
        return fail(YALPS_E_ARG, "yalps_tableau_set_shard: bad argument");
    if (t->height != 1 + bounds[rank + 1] - bounds[rank] || bounds[0] != 1 || bounds[nranks] != global_height)
        return fail(YALPS_E_ARG, "yalps_tableau_set_shard: uploaded rows do not match bounds (objective row + own rows)");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    Desc &d = t->d;
    d.nshards = nranks;
    d.shard_rank = rank;
    d.row_base = bounds[rank] - 1;
    for (int k = 0; k <= MAX_SHARDS; k++) d.bounds[k] = k <= nranks ? bounds[k] : INT_MAX;
    // the permutations are global (every rank replays the same basis swaps)
    const size_t n = (size_t)d.w + (size_t)global_height;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipFree(d.pos));
    HIP_TRY(hipFree(d.var));
    HIP_TRY(hipMalloc(&d.pos, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc(&d.var, sizeof(int32_t) * n));
    HIP_TRY(hipMemcpyAsync(d.pos, pos, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d.var, var, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    t->perm_len = (int32_t)n;
    d.perm_len = t->perm_len;
    t->rvar.fn = nullptr; // a shard is driven step by step
    // graphs captured for the unsharded tableau hold the old Desc
    for (int k = 0; k < 2; k++) {
        if (t->graph_exec[k]) (void)hipGraphExecDestroy(t->graph_exec[k]);
        if (t->graph[k]) (void)hipGraphDestroy(t->graph[k]);
        t->graph_exec[k] = nullptr;
        t->graph[k] = nullptr;
    }
    return 0;
}

int64_t yalps_shard_slot_doubles(const yalps_tableau *t) { return t ? SHARD_HDR + 2 * (int64_t)t->d.pitch : 0; }

int32_t yalps_shard_begin(yalps_tableau *t, double precision, double maxPivots) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_shard_begin: no tableau uploaded");
    HIP_TRY(hipSetDevice(t->ctx->device));
    int rc = init_state(t, precision, maxPivots, 0);
    if (rc) return rc;
    launch_one(t, 0, MODE_FUSED, 0); // bootstrap scan: emits this rank's first partials
    HIP_TRY(hipGetLastError());
    t->shard_parity = 1;
    return 0;
}

int32_t yalps_shard_select(yalps_tableau *t, double *send_dev) {
    if (!t || !send_dev) return fail(YALPS_E_ARG, "yalps_shard_select: bad argument");
    shard_select_kernel<<<dim3(1), dim3(1024), 0, t->ctx->stream>>>(t->d, t->shard_parity, send_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t yalps_shard_apply(yalps_tableau *t, const double *gathered_dev) {
    if (!t || !gathered_dev) return fail(YALPS_E_ARG, "yalps_shard_apply: bad argument");
    launch_one(t, t->shard_parity, MODE_SHARD, t->ctx->nt_stores ? 64 : 0, gathered_dev);
    HIP_TRY(hipGetLastError());
    t->shard_parity ^= 1;
    return 0;
}

int32_t yalps_shard_poll(yalps_tableau *t, int32_t *status_out, double *result_out, int64_t *pivots_out) {
    if (!t) return fail(YALPS_E_ARG, "yalps_shard_poll: bad argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    YState now;
    HIP_TRY(hipMemcpy(&now, t->d.st + t->shard_parity, sizeof(YState), hipMemcpyDeviceToHost));
    if (now.status != RUNNING) t->cur = now.mbuf;
    if (status_out) *status_out = now.status;
    if (result_out) *result_out = now.result;
    if (pivots_out) *pivots_out = now.pivots;
    return 0;
}

// ---- batched branch-and-cut node evaluation (BASELINE config 4) ---------------------------------
struct yalps_batch {
    yalps_ctx *ctx = nullptr;
    BatchDesc d{};
    int32_t max_nodes = 0, max_cuts = 0;
    double *root_mat = nullptr, *root_rhs = nullptr;
    int32_t *root_pos = nullptr, *root_var = nullptr;
    int32_t *cut_off = nullptr, *cut_sign = nullptr, *cut_var = nullptr;
    double *cut_val = nullptr;
    size_t shmem = 0;
    int32_t last_count = 0;
};

static int32_t batch_create_impl(yalps_ctx *ctx, int32_t width, int32_t root_height, int32_t max_cuts, int32_t max_nodes,
                                 yalps_batch **out) {
    HIP_TRY(hipSetDevice(ctx->device));
    yalps_batch *b = new yalps_batch();
    b->ctx = ctx;
    *out = b;
    b->max_nodes = max_nodes;
    b->max_cuts = max_cuts;
    BatchDesc &d = b->d;
    d.w = width;
    d.n = width - 1;
    d.pitch = (d.n + 15) / 16 * 16;
    d.h0 = root_height;
    d.hmax = root_height + max_cuts;
    d.permmax = width + d.hmax;
    b->shmem = sizeof(double) * ((size_t)d.hmax + d.pitch);
    if (b->shmem > 150 * 1024) return fail(YALPS_E_ARG, "yalps_batch_create: node tableau too large for the batched path");
    const size_t nm = (size_t)max_nodes;
    HIP_TRY(hipMalloc(&b->root_mat, sizeof(double) * (size_t)d.h0 * d.pitch));
    HIP_TRY(hipMemset(b->root_mat, 0, sizeof(double) * (size_t)d.h0 * d.pitch));
    HIP_TRY(hipMalloc(&b->root_rhs, sizeof(double) * (size_t)d.h0));
    HIP_TRY(hipMalloc(&b->root_pos, sizeof(int32_t) * (size_t)(width + d.h0)));
    HIP_TRY(hipMalloc(&b->root_var, sizeof(int32_t) * (size_t)(width + d.h0)));
    HIP_TRY(hipMalloc(&d.ws_mat, sizeof(double) * nm * d.hmax * d.pitch));
    HIP_TRY(hipMalloc(&d.ws_rhs, sizeof(double) * nm * d.hmax));
    HIP_TRY(hipMalloc(&d.ws_pos, sizeof(int32_t) * nm * d.permmax));
    HIP_TRY(hipMalloc(&d.ws_var, sizeof(int32_t) * nm * d.permmax));
    HIP_TRY(hipMalloc(&b->cut_off, sizeof(int32_t) * (nm + 1)));
    HIP_TRY(hipMalloc(&b->cut_sign, sizeof(int32_t) * nm * max_cuts));
    HIP_TRY(hipMalloc(&b->cut_var, sizeof(int32_t) * nm * max_cuts));
    HIP_TRY(hipMalloc(&b->cut_val, sizeof(double) * nm * max_cuts));
    HIP_TRY(hipMalloc(&d.status, sizeof(int32_t) * nm));
    HIP_TRY(hipMalloc(&d.height, sizeof(int32_t) * nm));
    HIP_TRY(hipMalloc(&d.result, sizeof(double) * nm));
    HIP_TRY(hipMalloc(&d.pivots, sizeof(long long) * nm));
    d.root_mat = b->root_mat;
    d.root_rhs = b->root_rhs;
    d.root_pos = b->root_pos;
    d.root_var = b->root_var;
    d.cut_off = b->cut_off;
    d.cut_sign = b->cut_sign;
    d.cut_var = b->cut_var;
    d.cut_val = b->cut_val;
    if (b->shmem > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(batch_kernel<256>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->shmem));
    *out = b;
    return 0;
}

int32_t yalps_batch_create(yalps_ctx *ctx, int32_t width, int32_t root_height, int32_t max_cuts, int32_t max_nodes,
                           yalps_batch **out) {
    if (!ctx || !out || width < 2 || root_height < 1 || max_cuts < 1 || max_nodes < 1)
        return fail(YALPS_E_ARG, "yalps_batch_create: bad argument");
    *out = nullptr;
    const int32_t rc = batch_create_impl(ctx, width, root_height, max_cuts, max_nodes, out);
    if (rc && *out) {
        const std::string why = g_err;
        yalps_batch_destroy(*out);
        *out = nullptr;
        g_err = why;
    }
    return rc;
}

void yalps_batch_destroy(yalps_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    void *bufs[] = {b->root_mat, b->root_rhs, b->root_pos, b->root_var, b->d.ws_mat, b->d.ws_rhs, b->d.ws_pos, b->d.ws_var,
                    b->cut_off, b->cut_sign, b->cut_var, b->cut_val, b->d.status, b->d.height, b->d.result, b->d.pivots};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete b;
}

int32_t yalps_batch_set_root(yalps_batch *b, const double *matrix, const int32_t *pos, const int32_t *var) {
    if (!b || !matrix || !pos || !var) return fail(YALPS_E_ARG, "yalps_batch_set_root: NULL argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t s = b->ctx->stream;
    const BatchDesc &d = b->d;
    HIP_TRY(hipMemcpy2DAsync(b->root_rhs, sizeof(double), matrix, sizeof(double) * d.w, sizeof(double), d.h0,
                             hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpy2DAsync(b->root_mat, sizeof(double) * d.pitch, matrix + 1, sizeof(double) * d.w,
                             sizeof(double) * d.n, d.h0, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->root_pos, pos, sizeof(int32_t) * (size_t)(d.w + d.h0), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->root_var, var, sizeof(int32_t) * (size_t)(d.w + d.h0), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int32_t yalps_batch_solve(yalps_batch *b, int32_t count, const int32_t *cut_offsets, const int32_t *cut_sign,
                          const int32_t *cut_var, const double *cut_value, double precision, double maxPivots,
                          int32_t *status_out, double *result_out, int64_t *pivots_out, float *gpu_ms_out) {
    if (!b || count < 1 || count > b->max_nodes || !cut_offsets || !cut_sign || !cut_var || !cut_value)
        return fail(YALPS_E_ARG, "yalps_batch_solve: bad argument");
    const int32_t total = cut_offsets[count];
    for (int32_t i = 0; i < count; i++)
        if (cut_offsets[i + 1] - cut_offsets[i] > b->max_cuts || cut_offsets[i + 1] < cut_offsets[i])
            return fail(YALPS_E_ARG, "yalps_batch_solve: a node has more cuts than the batch was created for");
    yalps_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b->cut_off, cut_offsets, sizeof(int32_t) * (size_t)(count + 1), hipMemcpyHostToDevice, s));
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(b->cut_sign, cut_sign, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->cut_var, cut_var, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->cut_val, cut_value, sizeof(double) * (size_t)total, hipMemcpyHostToDevice, s));
    }
    b->d.precision = precision;
    b->d.max_pivots = maxPivots;
    HIP_TRY(hipEventRecord(c->ev0, s));
    batch_kernel<256><<<dim3(count), dim3(256), b->shmem, s>>>(b->d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, s));
    if (status_out) HIP_TRY(hipMemcpyAsync(status_out, b->d.status, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost, s));
    if (result_out) HIP_TRY(hipMemcpyAsync(result_out, b->d.result, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    if (pivots_out) HIP_TRY(hipMemcpyAsync(pivots_out, b->d.pivots, sizeof(int64_t) * (size_t)count, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (gpu_ms_out) HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    b->last_count = count;
    return 0;
}

int32_t yalps_batch_download(yalps_batch *b, int32_t node, int32_t height, double *matrix, double *col0,
                             int32_t *pos, int32_t *var) {
    if (!b || node < 0 || node >= b->last_count || height < 1 || height > b->d.hmax)
        return fail(YALPS_E_ARG, "yalps_batch_download: bad argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t s = b->ctx->stream;
    const BatchDesc &d = b->d;
    const double *mat = d.ws_mat + (size_t)node * d.hmax * d.pitch;
    const double *rhs = d.ws_rhs + (size_t)node * d.hmax;
    if (matrix) {
        HIP_TRY(hipMemcpy2DAsync(matrix, sizeof(double) * d.w, rhs, sizeof(double), sizeof(double), height,
                                 hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpy2DAsync(matrix + 1, sizeof(double) * d.w, mat, sizeof(double) * d.pitch, sizeof(double) * d.n,
                                 height, hipMemcpyDeviceToHost, s));
    }
    if (col0) HIP_TRY(hipMemcpyAsync(col0, rhs, sizeof(double) * (size_t)height, hipMemcpyDeviceToHost, s));
    const size_t nperm = sizeof(int32_t) * (size_t)(d.w + height);
    if (pos) HIP_TRY(hipMemcpyAsync(pos, d.ws_pos + (size_t)node * d.permmax, nperm, hipMemcpyDeviceToHost, s));
    if (var) HIP_TRY(hipMemcpyAsync(var, d.ws_var + (size_t)node * d.permmax, nperm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
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
        rc = yalps_ctx_create(env_int("YALPS_HIP_DEVICE", 0), &g_default_ctx);
        if (rc) return rc;
    }
    yalps_tableau *t = g_default_tab;
    if (!t || t->d.w != width || t->d.hcap < height || t->d.hcap > 4 * height) {
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



