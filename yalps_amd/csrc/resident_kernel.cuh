// resident_kernel.cuh -- persistent kernel, tableau resident in the register files
// Part of libyalps_hip.so; included by the persistent_resident_*.hip translation units inside their unnamed namespaces
// (gfx950 only); its sc1 load / store helpers also serve stream_kernel.cuh.
#pragma once

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
// Two variations, template flags: X parks a few more rows per workgroup in LDS (tableaux a little beyond the register
// files); TAG sends the candidate row as self-validating granules (Guideline 16 R2) for narrow rows.
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
// one 16-byte record (flag polls: Guideline 16's table lists 16-B sc1 flag stores / polls)
__device__ __forceinline__ double2 ld16_sc1_one(const void *p) {
    v2f64 t;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory");
    return make_double2(t.x, t.y);
}
template <int J>
__device__ __forceinline__ void ld16_sc1(double2 (&out)[J], const double *base, const int (&ofs)[J]) {
    static_assert(J >= 1 && J <= 8, "lane units per row");
    if constexpr (J == 7) { // (all seven in flight behind one wait)
        v2f64 u[7];
        asm volatile("global_load_dwordx4 %0, %7, off sc1\n\tglobal_load_dwordx4 %1, %8, off sc1\n\t"
                     "global_load_dwordx4 %2, %9, off sc1\n\tglobal_load_dwordx4 %3, %10, off sc1\n\t"
                     "global_load_dwordx4 %4, %11, off sc1\n\tglobal_load_dwordx4 %5, %12, off sc1\n\t"
                     "global_load_dwordx4 %6, %13, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2]), "v"(base + ofs[3]), "v"(base + ofs[4]),
                       "v"(base + ofs[5]), "v"(base + ofs[6])
                     : "memory");
#pragma unroll
        for (int j = 0; j < 7; j++) out[j] = make_double2(u[j].x, u[j].y);
        return;
    }
    if constexpr (J == 8) { // two groups of four
        double2 lo[4], hi[4];
        const int olo[4] = {ofs[0], ofs[1], ofs[2], ofs[3]}, ohi[4] = {ofs[4], ofs[5], ofs[6], ofs[7]};
        ld16_sc1<4>(lo, base, olo);
        ld16_sc1<4>(hi, base, ohi);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            out[j] = lo[j];
            out[4 + j] = hi[j];
        }
        return;
    }
    v2f64 t[J > 6 ? 1 : J];
#define YLD "global_load_dwordx4 %"
    if constexpr (J == 1) {
        asm volatile(YLD "0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t[0]) : "v"(base + ofs[0]) : "memory");
    } else if constexpr (J == 2) {
        asm volatile(YLD "0, %2, off sc1\n\t" YLD "1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1])
                     : "v"(base + ofs[0]), "v"(base + ofs[1])
                     : "memory");
    } else if constexpr (J == 3) {
        asm volatile(YLD "0, %3, off sc1\n\t" YLD "1, %4, off sc1\n\t" YLD "2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2])
                     : "memory");
    } else if constexpr (J == 4) {
        asm volatile(YLD "0, %4, off sc1\n\t" YLD "1, %5, off sc1\n\t" YLD "2, %6, off sc1\n\t" YLD "3, %7, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2]), "v"(base + ofs[3])
                     : "memory");
    } else if constexpr (J == 5) {
        asm volatile(YLD "0, %5, off sc1\n\t" YLD "1, %6, off sc1\n\t" YLD "2, %7, off sc1\n\t" YLD "3, %8, off sc1\n\t"
                     YLD "4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2]), "v"(base + ofs[3]), "v"(base + ofs[4])
                     : "memory");
    } else {
        asm volatile(YLD "0, %6, off sc1\n\t" YLD "1, %7, off sc1\n\t" YLD "2, %8, off sc1\n\t" YLD "3, %9, off sc1\n\t"
                     YLD "4, %10, off sc1\n\t" YLD "5, %11, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5])
                     : "v"(base + ofs[0]), "v"(base + ofs[1]), "v"(base + ofs[2]), "v"(base + ofs[3]), "v"(base + ofs[4]),
                       "v"(base + ofs[5])
                     : "memory");
    }
#undef YLD
#pragma unroll
    for (int j = 0; j < J; j++) out[j] = make_double2(t[j].x, t[j].y);
}

// X = true: d.extra more rows per workgroup (slots R .. R + extra - 1, rows b + NB * slot like the others) are parked
// in LDS behind the basis, for tableaux a little beyond the register files: same arithmetic, same order of decisions,
// the rows are read and written with 16-byte ds accesses at the columns the lane also holds of the register rows.
constexpr int XROWS = 8; // most LDS rows per workgroup
// Stage stamps (CDNA guide 7, In-kernel stamps): only in the diagnostic build (-DYALPS_STAMPS -> libyalps_hip_stamps.so,
// tools/resident_stages.py); the shipped kernels execute none.  Sums of s_memtime differences per stage, kept in
// scalar registers, stored once by lane 0 when the launch ends, to a buffer nothing else reads.
#ifdef YALPS_STAMPS
#define YSTAMP(k)                                                                                           \
    do {                                                                                                    \
        unsigned long long t_;                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        st_acc[k] += t_ - st_last;                                                                          \
        st_last = t_;                                                                                       \
    } while (0)
#else
#define YSTAMP(k) do { } while (0)
#endif
#ifndef YALPS_SPLIT_NUM
#define YALPS_SPLIT_NUM 1 // quarters of the register rows eliminated while the candidate row's stores drain
#endif
// TAG = true (narrow rows, J <= 3): the candidate row travels as self-validating granules (Guideline 16 R2: "the data IS
// the flag"): every double is ONE 16-byte sc1 store of two 8-byte granules {epoch, low word} {epoch, high word}.  The
// publisher then neither drains nor joins a barrier before its key record leaves, and the readers of the winner's row
// re-read its granules until every tag carries the epoch -- one fabric round trip less per pivot where the pivot is
// all latency.  (The granule buffers are zeroed before every launch; epochs count from 1 within a launch.)
__device__ __forceinline__ double2 tagged(unsigned epoch, double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)epoch << 32;
    return make_double2(__longlong_as_double((long long)(t | (u & 0xffffffffull))), __longlong_as_double((long long)(t | (u >> 32))));
}
__device__ __forceinline__ bool untag(unsigned epoch, double2 g, double &v) {
    const unsigned long long lo = (unsigned long long)__double_as_longlong(g.x), hi = (unsigned long long)__double_as_longlong(g.y);
    v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
    return (unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch;
}
template <int T, int J, int R, bool X = false, bool TAG = false>
__global__ __launch_bounds__(T) void resident_kernel(Desc d, int parity, int chunk) {
    static_assert(!(X && TAG) && (!TAG || J <= 3), "tagged rows: narrow register-only variants");
    constexpr int SPLIT = TAG ? 0 : YALPS_SPLIT_NUM * R / 4; // other rows eliminated between the candidate row's stores and its flag
    __shared__ double sk[2][16];
    __shared__ int si[2][16];
    __shared__ double sh_val[R + 2]; // per-row broadcast: pivot-column entry / entering-column entry
    __shared__ double sh_nq[R + 2];  // -coef/quotient per row (:36), for the objective row, 1/quotient (:25)
    __shared__ double sh_ck;         // my candidate for the next exchange: key, row, local slot
    __shared__ int sh_ci, sh_cg, sh_fail, sh_flag, sh_verdict;
    __shared__ double sh_xcf[X ? XROWS : 1]; // LDS rows: pivot-column entry (kept for the whole pivot: no registers to hold it),
    __shared__ double sh_xnq[X ? XROWS : 1]; // -coef/quotient (:36),
    __shared__ double sh_xla[X ? XROWS : 1]; // entering-column entry for the candidate
    extern __shared__ int sh_perm[]; // workgroup 0: var[perm_len] then pos[perm_len]; X: then the parked rows

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
    int64_t hist_len = Sin->hist_len; // checkCycles: pivots recorded in the current phase (src/simplex.ts:67,107)
    const bool check_cycles = C->check_cycles != 0;
    int slot = 0;

    int cofs[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c0 = 2 * (tid + j * T);
        cofs[j] = c0 < pitch ? c0 : 0;
    }
    const int E = X ? d.extra : 0;
    double *const xl = reinterpret_cast<double *>(sh_perm + (X ? d.xl_ofs : 0)); // row slot R + e at xl + e * pitch
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
    if constexpr (X) {
        for (int e = 0; e < E; e++) {
            const int r = b + NB * (R + e);
            const double *mr = matA + (size_t)(r < h ? r : b) * pitch;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c0 = 2 * (tid + j * T);
                if (c0 < pitch) *reinterpret_cast<double2 *>(xl + e * pitch + c0) = *reinterpret_cast<const double2 *>(mr + c0);
            }
        }
    }
    const int my_r = b + NB * tid; // lane g = tid < R (+ E) owns the scalar side of row slot g
    const bool my_live = tid < R + E && my_r < h;
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
    unsigned epoch = 0; // exchange round: publish() opens round epoch + 1 for the candidate candidate() has just left
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
                const double value = (X && tid >= R) ? sh_xla[tid - R] : sh_val[tid];
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
                if constexpr (TAG) // the key record needs nothing else: it leaves before the candidate row is even finished
                    st16_sc1(reinterpret_cast<double *>(d.rc_flag[(epoch + 1) & 1] + 2 * b),
                             make_double2(c.k, __longlong_as_double((long long)(((unsigned long long)(epoch + 1) << 32) | (unsigned)c.i))));
            }
        }
        __syncthreads();
    };
    // publish my candidate (sh_ck / sh_ci) and the data of its row (register slot sh_cg), in two steps: the stores, and --
    // once they have drained -- the flag.  Between the two the caller eliminates SPLIT of its other rows: the drain is
    // 1-2 us in which the workgroup would otherwise do nothing.
    auto publish_stores = [&]() __attribute__((always_inline)) {
        epoch++;
        const int par = epoch & 1, cg = sh_cg;
        if constexpr (TAG) { // (the key record of this epoch left inside candidate())
            double *dst = d.rc_tag[par] + (size_t)b * (2 * pitch + 2);
#pragma unroll
            for (int g = 0; g < R; g++) {
                if (g == cg) {
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        const int c0 = 2 * (tid + j * T);
                        if (c0 < pitch) {
                            st16_sc1(dst + 2 * c0, tagged(epoch, x[g][j].x));
                            st16_sc1(dst + 2 * c0 + 2, tagged(epoch, x[g][j].y));
                        }
                    }
                }
            }
            if (tid == cg) st16_sc1(dst + 2 * pitch, tagged(epoch, my_rhs)); // the candidate row's RHS entry (lane cg)
            return; // (no drain, no barrier, no flag: every granule says for itself which epoch it belongs to)
        }
        // x[cg] straight from its registers: one uniform branch per slot, the slot index stays a
        // compile-time constant (a value select over the slots cost 4 R J v_cndmask and, with the copies
        // hipcc made for it, twice the registers; a runtime index would move the rows to scratch)
        double *dst = d.rc_rows[par] + (size_t)b * pitch;
#pragma unroll
        for (int g = 0; g < R; g++) {
            if (g == cg) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 < pitch) st16_sc1(dst + c0, x[g][j]);
                }
            }
        }
        if constexpr (X) {
            if (cg >= R) { // the candidate row is a parked one
                const double *src = xl + (cg - R) * pitch;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 < pitch) st16_sc1(dst + c0, *reinterpret_cast<const double2 *>(src + c0));
                }
            }
        }
        if (tid == cg) st_sc1(d.rc_key[par] + b, my_rhs); // the candidate row's RHS entry (lane cg)
    };
    auto publish_flag = [&]() __attribute__((always_inline)) {
        if constexpr (TAG) return;
        const int par = epoch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
        __syncthreads();                                  // ... before ONE lane raises the flag:
        if (tid == 0) // ONE 16-byte record {candidate key, epoch << 32 | row}, one store, polled with one 16-byte load
            st16_sc1(reinterpret_cast<double *>(d.rc_flag[par] + 2 * b),
                     make_double2(sh_ck, __longlong_as_double((long long)(((unsigned long long)epoch << 32) | (unsigned)sh_ci))));
    };
    auto publish = [&]() __attribute__((always_inline)) {
        publish_stores();
        publish_flag();
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
        if constexpr (X)
            if (la > 0 && tid < E) sh_xla[tid] = xl[tid * pitch + la - 1];
        __syncthreads();
    };
    int done = 0, term = RUNNING;
    double term_result = NAN;
    bool stop = false;
#ifdef YALPS_STAMPS
    unsigned long long st_acc[20] = {}, st_last = 0, st_t0 = 0, st_r0 = 0;
#endif
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
#ifdef YALPS_STAMPS
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0)::"memory");
    st_last = st_t0;
#endif
    // (single back edge, single exit: every `stop` is a flag, so the rows stay in one set of registers)
    while (!stop) {
        // ---------------- gather everyone's candidate -------------------------------------------
        const int par = epoch & 1;
        KI c = {INFINITY, INT_MAX};
        if (tid < NB) {
            // key and tag are one 16-byte record, written by one store and read by one load
            unsigned long long f = 0;
            unsigned spins = 0;
            [[maybe_unused]] unsigned long long spin_t0 = 0;
            double2 rec;
            for (;;) {
                rec = ld16_sc1_one(d.rc_flag[par] + 2 * tid);
                f = (unsigned long long)__double_as_longlong(rec.y);
                if ((unsigned)(f >> 32) == epoch) break;
                if (RESIDENT_SPIN_EXPIRED(spins, spin_t0, d.rc_err)) {
                    sh_fail = 1;
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            c.i = (int)(unsigned)f;
            c.k = rec.x;
#ifdef YALPS_AB_POLL_GUARD
            // (never expected: a record that names no row of this tableau -- leave through the failure exit, with the error
            // word set, instead of indexing with it)
            if (c.i != INT_MAX && (unsigned)c.i >= (unsigned)h) {
                sh_fail = 1;
                __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
        }
        YSTAMP(0); // wait for everybody's flag (waves 0 .. NB/64 - 1; the others go straight to the barrier)
        c = block_argmin<T>(c, sk, si, slot); // (its barrier is the one the polling waves join)
        slot ^= 1;
        YSTAMP(1);
        if (sh_fail) return; // uniform: written before the barrier above
        if (c.i == INT_MAX) {
            if (phase == 1) { // :120 phase 1 is over: same tableau, now the min-ratio exchange
                phase = 2;
                iter = 0.0;
                hist_len = 0;
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
            YSTAMP(14);
            continue;
        }
#if defined(YALPS_AB_RETURN_GUARD) || defined(YALPS_AB_NOGUARD) || defined(YALPS_AB_POLL_GUARD)
        const int row = c.i, owner = row % NB;
#else
        // (never expected: a record that names no row of this tableau.  No exit of its own -- a `return` here, or a test in
        // the poll loop, cost the tall variants 4-11 % per pivot through register allocation alone, same-box A/B --: the
        // index is clamped, so that nothing is addressed with it, and the error word makes the host discard the launch)
        const int row_in = c.i, row = (unsigned)row_in < (unsigned)h ? row_in : 0, owner = row % NB;
        if (row != row_in && tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#ifdef YALPS_AB_RETURN_GUARD
        if ((unsigned)row >= (unsigned)h) { // (never expected: a record that names no row of this tableau -- leave with the error
            if (tid == 0) __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // word set instead of indexing with it)
            return;
        }
#endif
        // ---------------- the winner's raw row (sc1 loads only) ----------------------------------
        double rhs_row;
        double2 pv[J];
        if constexpr (TAG) {
            const double *src = d.rc_tag[par] + (size_t)owner * (2 * pitch + 2);
            int gofs[2 * J + 1];
#pragma unroll
            for (int j = 0; j < J; j++) {
                gofs[2 * j] = 2 * cofs[j];
                gofs[2 * j + 1] = 2 * cofs[j] + 2;
            }
            gofs[2 * J] = 2 * pitch;
            unsigned spins = 0;
            [[maybe_unused]] unsigned long long spin_t0 = 0;
            for (;;) { // every wave for itself: re-read my granules until each carries this epoch
                double2 g[2 * J + 1];
                ld16_sc1<2 * J + 1>(g, src, gofs);
                bool ok = untag(epoch, g[2 * J], rhs_row);
#pragma unroll
                for (int j = 0; j < J; j++) {
                    ok &= untag(epoch, g[2 * j], pv[j].x);
                    ok &= untag(epoch, g[2 * j + 1], pv[j].y);
                }
                if (__all(ok)) break;
                if (RESIDENT_SPIN_EXPIRED(spins, spin_t0, d.rc_err)) {
                    sh_fail = 1; // (acted upon behind the next barrier, where every wave sees it)
                    __hip_atomic_store(d.rc_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            const double *src = d.rc_rows[par] + (size_t)owner * pitch;
            rhs_row = ld_sc1(d.rc_key[par] + owner);
            ld16_sc1<J>(pv, src, cofs);
        }
        YSTAMP(2); // the winner's row
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
            if (TAG && sh_fail) return; // (the wait for the winner's granules gave up)
            if (e.i == INT_MAX) { // :135
                term = YALPS_INFEASIBLE;
                stop = true;
                continue;
            }
            col = e.i;
        }
        if (check_cycles) { // :98,137 hasCycle before the pivot: workgroup 0 (it holds the basis) decides for everybody
            int cycled = 0;
            if (b == 0) {
                const int leaving = sh_perm[w + row], entering = sh_perm[col]; // var[] = sh_perm[0 .. perm_len)
                cycled = has_cycle(C, hist_len, leaving, entering, &sh_flag) ? 1 : 0;
                if (tid == 0)
                    __hip_atomic_store(d.rc_verdict + par, ((unsigned long long)epoch << 32) | (unsigned)cycled,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) {
                    unsigned long long v = 0;
                    unsigned spins = 0;
            [[maybe_unused]] unsigned long long spin_t0 = 0;
                    for (;;) {
                        v = __hip_atomic_load(d.rc_verdict + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(v >> 32) == epoch) break;
                        if (RESIDENT_SPIN_EXPIRED(spins, spin_t0, d.rc_err)) {
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
            if (cycled) { // ["cycled", NaN]: the tableau stays as it was before this pivot
                term = YALPS_CYCLED;
                stop = true;
                continue;
            }
        }
        YSTAMP(3); // phase 1: entering column; checkCycles: verdict
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
        if constexpr (X)
            if (tid < E) sh_xcf[tid] = xl[tid * pitch + col - 1]; // (the rows are complete: every wave has passed the gather's barrier)
        __syncthreads();
        YSTAMP(4); // pivot column of my rows through LDS
        if (TAG && sh_fail) return; // (the wait for the winner's granules gave up: uniform behind this barrier)
        const double q = sh_val[R + 1], coef0 = sh_val[R];
        double cf[R]; // uniform: pivot-column entry of each of my rows
#pragma unroll
        for (int g = 0; g < R; g++) // (scalar registers where the VGPR budget of 2 waves per SIMD is short: measured
            cf[g] = (T >= 512 && J * R >= 24) ? uniform_f64(sh_val[g]) : sh_val[g]; // +5 % at <512,2,9>, -10 % at <512,3,9>)
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
        if constexpr (X)
            if (tid >= R + 2 && tid < R + 2 + E) sh_xnq[tid - (R + 2)] = -sh_xcf[tid - (R + 2)] / q;
        if (my_live) { // RHS entry of my row (:33 at column 0)
            const double pn_rhs = nz_rhs ? rhs_row / q : 0.0;
            double my_coef = 0.0;
#pragma unroll
            for (int g = 0; g < R; g++)
                if (tid == g) my_coef = cf[g];
            if constexpr (X)
                if (tid >= R) my_coef = sh_xcf[tid - R];
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
        YSTAMP(5); // normalise, column divisions, RHS, objective replica
        // my rows, fully, as pivot() leaves them: slot `only` (only_it = true) or all slots but it
        // (register slots [glo, ghi); the rows parked in LDS go with the call that ends at R)
        auto finish_rows = [&](int only, bool only_it, int glo, int ghi) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < R; g++) { // (g must stay a compile-time index: the rows are registers)
            if (g < glo || g >= ghi) continue;
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
            if constexpr (X) {
                for (int e = 0; e < (ghi == R ? E : 0); e++) { // the parked rows: the same, through LDS
                    const int g = R + e;
                    if ((g == only) != only_it) continue;
                    double *xr = xl + e * pitch;
                    const double cfe = sh_xcf[e];
                    if (g == lslot) {
#pragma unroll
                        for (int j = 0; j < J; j++) {
                            const int c0 = 2 * (tid + j * T);
                            double2 v = pv[j];
                            if (tid == col_tid && j == col_j) v = with_elem(v, ecol, sh_nq[R + 1]); // :25
                            if (c0 < pitch) *reinterpret_cast<double2 *>(xr + c0) = v;
                        }
                    } else if (b + NB * g < h && fabs(cfe) > 1e-16) { // :31
#pragma unroll
                        for (int j = 0; j < J; j++) {
                            const int c0 = 2 * (tid + j * T);
                            if (c0 < pitch) {
                                double2 v = *reinterpret_cast<const double2 *>(xr + c0);
                                const double px = cfe * pv[j].x, py = cfe * pv[j].y;
                                const double nx = v.x - px, ny = v.y - py;
                                v.x = (nzmask & (1u << (2 * j))) ? nx : v.x;
                                v.y = (nzmask & (1u << (2 * j + 1))) ? ny : v.y;
                                if (tid == col_tid && j == col_j) v = with_elem(v, ecol, sh_xnq[e]); // :36
                                *reinterpret_cast<double2 *>(xr + c0) = v;
                            }
                        }
                    }
                }
            }
        };
        iter += 1.0;
        pivots += 1;
        done += 1;
        price(); // la of the next pivot, from the updated objective replica
        check();
        YSTAMP(6);
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
                            if constexpr (X) {
                                for (int e = 0; e < E; e++) { // (the parked rows still hold their entries from before this pivot)
                                    double v = xl[e * pitch + la - 1];
                                    const double cfe = sh_xcf[e];
                                    if (R + e == lslot)
                                        v = la == col ? sh_nq[R + 1] : p;
                                    else if (b + NB * (R + e) < h && fabs(cfe) > 1e-16) {
                                        if (la == col)
                                            v = sh_xnq[e];
                                        else if (nz) {
                                            const double prod = cfe * p;
                                            v = v - prod;
                                        }
                                    }
                                    sh_xla[e] = v;
                                }
                            }
                        }
                }
                __syncthreads();
            }
            YSTAMP(7); // my rows' entries of the next entering column
            candidate(phase);
            YSTAMP(8);
            const int cg = sh_cg;
            finish_rows(cg, true, 0, R);
            YSTAMP(9);
            publish_stores();
            YSTAMP(10);
            if constexpr (SPLIT > 0) finish_rows(cg, false, 0, SPLIT); // (while the stores drain)
            YSTAMP(11);
            publish_flag();
            YSTAMP(12);
            finish_rows(cg, false, SPLIT, R); // (while the flags travel)
        } else {
            finish_rows(-1, false, 0, R);
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
        YSTAMP(13); // the other rows
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
    if constexpr (X) {
        for (int e = 0; e < E; e++) {
            const int r = b + NB * (R + e);
            if (r < h) {
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c0 = 2 * (tid + j * T);
                    if (c0 < pitch) *reinterpret_cast<double2 *>(matB + (size_t)r * pitch + c0) = *reinterpret_cast<const double2 *>(xl + e * pitch + c0);
                }
            }
        }
    }
    if (my_live) rhsB[my_r] = my_rhs;
    if (b == 0) {
        // (the last pivot's basis swap was one lane's LDS writes at the very end of the loop body, with no barrier behind
        // them: a wave that got here first copied the old entries -- seen once in 380 GPU tests, on a 60-pivot solve)
        __syncthreads();
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
            Sout->hist_len = hist_len;
            Sout->iter = iter;
            Sout->result = term_result;
            Sout->pivots = pivots;
        }
    }
}
