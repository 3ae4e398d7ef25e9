// common.cuh -- device-side state, descriptors, DPP arg-min reductions and small helpers shared by all kernels
// Part of libyalps_hip.so; included by yalps_hip.hip inside its anonymous namespace (gfx950 only).
#pragma once


constexpr int RUNNING = -1;
constexpr int SHARD_SWEEP_MISSING = 77; // internal: a shard's step kernel found `depth` pivots still pending -- the sweep launch (dsweep_kernel.cuh) did not run; yalps_shard_run fails on it
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

// Row shards with delayed row updates (dshard_kernel.cuh): what is pending between two launches, ping-ponged like YState.
struct alignas(16) DelayState {
    int32_t npend;     // pivots decided whose eliminations have not been carried out on this rank's rows
    int32_t lav_valid; // d.dlav holds my rows' entries of column YState::la as they are now
    int32_t pad_[2];
    int32_t pl[16]; // per pending pivot, oldest first: the pivot row as a LOCAL row of this rank (-1: another rank's)
    int32_t pc[16]; // ... its pivot column (mat index)
};
// rows in flight per wave in the panel sweep (panel_flush.cuh): YALPS_PANEL_SETS register sets of YALPS_PANEL_D rows of 8 units
// (same-box A/B, 16385 columns: two rows sharing every read of a pending row's units -- D 2, one set -- against one row in each of
// two sets: stream3 16385^2 9.06 -> 8.60 s per whole solve, dshard 16385 rows 101.8 -> 97.4 us per pivot, 8193 rows 73.8 -> 71.0;
// two sets of two rows: no better, 203 registers)
#ifndef YALPS_PANEL_D
#define YALPS_PANEL_D 2
#endif
#ifndef YALPS_PANEL_SETS
#define YALPS_PANEL_SETS 1
#endif
constexpr int HP_SCAL = 8 + 2 * 16; // doubles a workgroup publishes with its candidate's key in stream3_kernel's two-step exchange: [0] the row's
                                    // RHS entry, [1] bit p: it was pending pivot p's pivot row, [8 + p] its entry of p's column as it was, [8 + 16 + p] what replaces it
constexpr int STREAM3_MAXD = 16;         // pending pivots stream3_kernel can hold (the depth in use is Desc::delay_depth)
constexpr int STREAM3_PANEL_UNITS = 512; // 16-byte units of a row per LDS panel of its sweep (panel_flush.cuh): 1024 columns
// (Tried for rows of 16 units per lane: panels of 384 units and up to 22 pending pivots -- what the LDS of a CU holds that way.  Same
// box, whole solves at 16385^2: 384 units / 16 pending 80.3 us per pivot, / 20 76.3, / 22 74.6, against 76.8 with 512 units / 16; at
// 4097 x 16385 and 3001 x 16385 every 384-unit form lost 1-2 us.  A pending pivot more costs a sweep what its arithmetic costs --
// 29 us at 16385^2 --: the sweep is the SUM of the rows' memory time and the arithmetic, not the larger of the two.)
__host__ __device__ constexpr int stream3_panel_units(int) { return STREAM3_PANEL_UNITS; }
constexpr int STREAM3_DEPTH_J16 = 16;    // pending pivots of the 16-unit form by default
constexpr int DSHARD_MAXD = 16;          // pending pivots a row shard can hold (dshard_kernel.cuh); the depth in use is Desc::delay_depth
constexpr int DSHARD_PANEL_UNITS = 512;  // 16-byte units of a row per LDS panel of its sweep (panel_flush.cuh): 1024 columns

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
    double *rc_tag[2];              // [nb][2 * pitch + 2] the same as self-validating granules (resident_kernel<.., TAG>)
    unsigned long long *rc_flag[2]; // [nb][2] {candidate key bits, (epoch << 32) | global row index}
    int32_t *rc_err;                // set when a workgroup gives up waiting (never expected)
    unsigned long long *rc_verdict; // [2] checkCycles: workgroup 0's verdict on the pivot of an epoch, (epoch << 32) | cycled
    unsigned long long *rc_rowflag; // [2][2] (round 2's two-step exchange of stream3_kernel; unused since the winner's row is materialised by its owner's XCD)
    // stream3_kernel's two-step exchange (round 3): [nb] 1 + XCD of every workgroup; [2][nb] "my slice of the winner's row is out" flags
    // (the epoch); [2][nb][HP_SCAL] the scalars every workgroup publishes with its candidate's key -- all in the zeroed control block
    double *ob_park; // stream3_kernel: [nb][pitch] where a workgroup parks its objective replica (registers) while it sweeps its rows
    int32_t *hp_xcc;
    unsigned long long *hp_flag;
    double *hp_scal;
    int32_t perm_len;
    int32_t extra;  // resident_kernel<.., true>: rows per workgroup parked in LDS (0: none)
    int32_t xl_ofs; // ... and where they start in the dynamic LDS block, in int32 units (behind var[] / pos[] at capacity)
    // any-shape fallback (generic_kernels.cuh): the normalised pivot row [pitch] and {quotient, its RHS, RHS non-zero}
    double *gen_prow, *gen_scal;
    // sweep_kernel (persistent, in place, rows of 8194 .. 16385 columns): its records in the zeroed control block, and
    // whether the row traffic is non-temporal (tableau beyond the Infinity Cache)
    unsigned long long *sw_sync, *sw_recs;
    int32_t sw_nt;
    // stream2_kernel: pivots whose eliminations are delayed and carried out together (as many pivot rows as LDS holds, <= 8)
    int32_t delay_depth;
    double *pend; // stream3_kernel: [8 XCDs][2 sets][delay_depth][pitch] the pending normalised pivot rows (one scratch per XCD)
    // row shards swept IN PLACE (wide_kernel<.., true, ..>): the objective row is the one row every workgroup reads while
    // its owner rewrites it, so it alone stays ping-ponged, as two replicas [pitch] beside the tableau
    double *obj[2];
    // row shards with delayed row updates (dshard_kernel.cuh; d.delay_depth pivots per sweep): the pending normalised pivot
    // rows [depth][pitch], my rows' entries of their pivot columns and what replaces them [depth][hcap] each, my rows'
    // entries of the next entering column [hcap], the pending pivots' rows / columns [2] by launch parity
    double *dpend, *dcolv, *dnqv, *dlav;
    DelayState *dstate;
    // row shards with checkCycles: shard_cycle_kernel's verdict on the pivot the step launch of the same parity is about to
    // carry out ([2] by launch parity; 1 = hasCycle, src/simplex.ts:98,137)
    int32_t *cyc_verdict;
    // row shards with delayed row updates: 1 = the step kernel leaves `delay_depth` pivots pending and the host enqueues
    // dshard_sweep_kernel (dsweep_kernel.cuh) behind it; 0 = the step kernel sweeps its rows itself
    int32_t ext_sweep;
    // diagnostic build only (-DYALPS_STAMPS, never the shipped library): [nb][STAMP_WORDS] per-workgroup stage sums in
    // shader cycles, written once when a persistent launch ends; no kernel reads it
    unsigned long long *dbg;
};
constexpr int STAMP_WORDS = 24; // stage sums [0..19], pivots [20], s_memtime span [21], s_memrealtime span [22]

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
// a value every lane of the wave holds, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
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

// Bounded waits of the persistent kernels' hand-offs.  A poll loop calls this once per unsuccessful poll; every 64th call
// reads the 100 MHz real-time counter and the launch's error word: true = give up (somebody else already has, or this
// wait has lasted SPIN_GIVE_UP_TICKS: the grid is not co-resident -- a foreign kernel holds CUs -- or a workgroup died).
// The caller then sets the error word and leaves; the host falls back to the launch-per-pivot kernels (yalps_hip.hip).
constexpr unsigned long long SPIN_GIVE_UP_TICKS = 5000000ull; // 50 ms (a pivot's hand-off takes microseconds)
// The register-resident kernels bound the same wait by a poll count instead: 2^16 polls of one sc1 round trip + s_sleep 2
// each (0.8-1.5 us) = 50-100 ms.  (Same-box A/B: reading the clock in their poll loops, and a `return` of its own for the
// row range check, cost the tall generation-1 variants 4-11 % per pivot -- 10001 x 1001: 14.7 -> 16.4 us -- through
// register allocation alone; the range check now rides on the poll loop's existing failure exit.)
constexpr unsigned SPIN_GIVE_UP_POLLS = 1u << 16;
#ifdef YALPS_AB_TIME_SPIN
#define RESIDENT_SPIN_EXPIRED(spins, t0, err) spin_expired(spins, t0, err)
#else
#define RESIDENT_SPIN_EXPIRED(spins, t0, err) \
    (++(spins) > SPIN_GIVE_UP_POLLS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
#endif
__device__ __forceinline__ bool spin_expired(unsigned &spins, unsigned long long &t0, const int32_t *err_word) {
    if ((++spins & 63u) != 0) return false;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (spins == 64u) t0 = now;
    return now - t0 > SPIN_GIVE_UP_TICKS || __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
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

// 16-byte row load / store; nt = non-temporal (streaming) cache policy, one dwordx4 instruction
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_row(const double *p, bool nt) {
    if (nt) {
        const v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const v2f64 *>(p));
        return make_double2(v.x, v.y);
    }
    return *reinterpret_cast<const double2 *>(p);
}
__device__ __forceinline__ void st_row_nt(double *p, double2 v) {
    v2f64 t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2f64 *>(p));
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
