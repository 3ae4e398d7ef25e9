// yalps_hip.hip -- MI355X (gfx950 / CDNA4) dense-tableau simplex core: host side + C ABI.
//
// Replaces the body of the reference's `simplex` export (src/simplex.ts:106-144) behind the C ABI
// of include/yalps_hip.h.  Written for gfx950 only.  DESIGN.md has the full picture; the device
// code lives in the .cuh files included below:
//   common.cuh           state / descriptors, DPP (key,index) arg-min reductions, small helpers
//   resident_kernel.cuh  the fast tier: persistent kernel, tableau resident in the register files,
//                        one L2 exchange (candidates + candidate rows) per pivot   [compiled in persistent_resident_*.hip]
//   pivot_kernel.cuh     streaming, one launch per pivot, rows batched in registers, tableau
//                        ping-ponged in HBM; also DECIDE/APPLY for checkCycles
//   wide_kernel.cuh      streaming for tableaux too wide / tall for register batches (pivot row in LDS)
//   shard_kernels.cuh    row-sharded solve across GPUs: per-rank select kernel (+ MODE_SHARD above)
//   assemble_kernels.cuh initial tableau from its written cells; applyCuts (branch-and-cut nodes) in HBM
//   stream_kernel.cuh    persistent in-place pivot loop for tableaux beyond the on-chip size   [compiled in persistent_stream.hip]
//   generic_kernels.cuh  any-shape fallback (rows wider than 16385 columns): DECIDE + APPLY launch per pivot
//   wg_simplex.cuh       the whole simplex loop by one workgroup; small_kernel (tableau in LDS)
//   batch_kernel.cuh     batched branch-and-cut nodes, one workgroup per node
//   milp_host.inc        (host C++) the whole branch and cut in one native call: yalps_milp_f64
// Host side here: contexts, tableaux (HBM layout, upload/download/assemble/apply_cuts), the solve driver
// (one workgroup in LDS | persistent chunks: register-resident, then in place | hipGraph batches of 64
// launches polled once per batch -- never a host round trip per pivot), shard steps, node batches, and
// the host-array drop-ins yalps_simplex_f64 / yalps_simplex_sparse_f64.
//
// Bit-exactness contract (tests/ compare against the oracle bit for bit): separately rounded
// multiply and subtract (-ffp-contract=off, checked in the ISA: v_mul_f64 + v_add_f64, no
// v_fma_f64 in the update), IEEE division, the 1e-16 flush / skip rules of pivot(), strict
// first-wins comparisons in every scan.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/yalps_hip.h"
#include "persistent_tables.h"

#pragma clang fp contract(off)

namespace {

#include "common.cuh"
#include "pivot_kernel.cuh"
#include "wide_kernel.cuh"
#include "shard_kernels.cuh"
#include "assemble_kernels.cuh"
#include "generic_kernels.cuh"
#include "wg_simplex.cuh"
#include "batch_kernel.cuh"

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

// ... and the in-place forms for row shards ([0] plain, [1] non-temporal row loads)
struct WInplace {
    int T, J; // the ping-pong variant it stands in for
    KernelFn fn[2];
    int launchT; // lanes of the in-place form (16 385-column rows: 512 x 16 units -- two 64-register row buffers fit the 256
                 // registers of a 512-lane workgroup; <1024,8> has 128 and spills 17 of them)
};
const WInplace kWideInplace[] = {
    {256, 1, {wide_kernel<256, 1, true, false>, wide_kernel<256, 1, true, true>}, 256},
    {256, 2, {wide_kernel<256, 2, true, false>, wide_kernel<256, 2, true, true>}, 256},
    {1024, 1, {wide_kernel<1024, 1, true, false>, wide_kernel<1024, 1, true, true>}, 1024},
    {1024, 2, {wide_kernel<1024, 2, true, false>, wide_kernel<1024, 2, true, true>}, 1024},
    {1024, 4, {wide_kernel<1024, 4, true, false>, wide_kernel<1024, 4, true, true>}, 1024},
    {1024, 8, {wide_kernel<512, 16, true, false>, wide_kernel<512, 16, true, true>}, 512},
};

using ResidentFn = void (*)(Desc, int, int);
struct RVariant {
    int T, J, R;
    ResidentFn fn;
};
// The persistent kernels' instantiations live in translation units of their own (persistent_*.hip, compiled side by
// side; persistent_tables.h).  A tableau takes the feasible resident variant (T * J 16-byte units span a row, R rows
// per workgroup) with the fewest row registers J * R -- few, fat lanes: the loop is latency-bound, and <= 8 waves per
// CU leave each lane 256 VGPRs; J = 3 / 5 keep 1025..1536 / 2049..2560 units out of the next power of two.
std::vector<RVariant> variants_of(std::initializer_list<PersistentTable> tables) {
    std::vector<RVariant> out;
    for (const PersistentTable &t : tables)
        for (int i = 0; i < t.count; i++)
            out.push_back({t.entries[i].T, t.entries[i].J, t.entries[i].R,
                           reinterpret_cast<ResidentFn>(const_cast<void *>(t.entries[i].fn))});
    return out;
}
const std::vector<RVariant> kResident = variants_of({yalps_resident_table_a(), yalps_resident_table_b()});
// resident2_kernel<T, J, R>: the same shapes with the second-generation pivot loop (resident2_kernel.cuh); used where it
// exists unless YALPS_HIP_RESIDENT_GEN=1 asks for the first generation (A/B measurements, tests of both)
const std::vector<RVariant> kResident2 = variants_of({yalps_resident2_table_a(), yalps_resident2_table_b()});
// The same with up to XROWS more rows per workgroup parked in LDS (tableaux a little beyond the register files):
// tried when no variant above fits; R = register rows, the LDS rows are what is missing.
const std::vector<RVariant> kResidentLds = variants_of({yalps_resident_lds_table()});
// resident_kernel<T, J, R, false, true>: the candidate row as self-validating granules (narrow rows; resident_kernel.cuh)
const std::vector<RVariant> kResidentTag = variants_of({yalps_resident_tag_table()});
constexpr int XROWS = YALPS_RESIDENT_LDS_MAX_ROWS;
// stream_kernel<lanes, 16-byte units per lane and row, hasCycle>: same signature as the resident kernel
const std::vector<RVariant> kStream = variants_of({yalps_stream_table()});
const std::vector<RVariant> kStreamCheck = variants_of({yalps_stream_check_table()});
// sweep_kernel<lanes, units, hasCycle>: persistent, in place, for what streams from HBM (rows of 8194 .. 16385 columns, and
// 4098 .. 8193-column tableaux beyond the Infinity Cache)
const std::vector<RVariant> kSweep = variants_of({yalps_sweep_table()});
const std::vector<RVariant> kSweepCheck = variants_of({yalps_sweep_check_table()});
const std::vector<RVariant> kStream2 = variants_of({yalps_stream2_table()}); // (R = non-temporal row traffic)
const std::vector<RVariant> kStream3 = variants_of({yalps_stream3_table(), yalps_stream3d_table()});
const std::vector<RVariant> kStream3Check = variants_of({yalps_stream3_check_table(), yalps_stream3d_check_table()});
const std::vector<RVariant> kDshard = variants_of({yalps_dshard_table()}); // (launch-per-pivot: (Desc, parity, mode, force, gather); R = non-temporal row traffic)
constexpr size_t SWEEP_BEYOND_CACHE = 200u << 20; // tableau bytes from which row traffic goes non-temporal (Infinity Cache: 256 MiB)
constexpr int STREAM3_PANEL_MIN_ROWS = 12; // rows per workgroup from which stream3_kernel's sweep goes through LDS panels (panel_flush.cuh)
constexpr int STREAM3_DEFAULT_DEPTH_WIDE = 16; // pending pivots of stream3_kernel for rows of 4098+ columns with 8+ rows per workgroup (YALPS_HIP_DELAY_DEPTH)
constexpr int DSHARD_XSWEEP_BELOW_ROWS = 24;   // rows per workgroup below which a row shard's sweep is a launch of its own (dsweep_kernel.cuh)
constexpr int DSHARD_PANEL_MIN_ROWS = 12;      // rows per workgroup from which a row shard's sweep goes through LDS panels (panel_flush.cuh)
constexpr int DSHARD_DEFAULT_DEPTH_PANEL = 16; // ... and its pending pivots then
constexpr int DSHARD_DEFAULT_DEPTH = 8; // pending pivots of a row shard (YALPS_HIP_DELAY_DEPTH, at most DSHARD_MAXD = 16)
constexpr int RESIDENT_CHUNK = 4096; // pivots per launch of the resident kernel (bounds its run time)

struct YalpsNcclId { // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
    char internal[128];
};

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// Dynamic LDS beyond 48 KB needs the function attribute raised first.  The attribute belongs to the FUNCTION (per device),
// not to a launch: raised once to the most a workgroup can have (one workgroup per CU; every kernel's static part is
// below 2 KB), never to the size one tableau happens to need -- a second tableau with a smaller need would lower it
// under the first one's feet.
std::mutex &persistent_mutex(int device) {
    static std::mutex mu[64];
    return mu[device & 63];
}
constexpr int PERSISTENT_RETRY_AFTER = 8;
// The same between processes: an advisory lock on one file per physical device (named by its PCI bus id), held from a
// persistent launch to its completion.  Two processes of this library on one GPU (ranks rehearsing on a shared card, a
// Node process next to a Python one) then never interleave their grids.  Kernels of other software are not covered:
// against those the launch has its bounded waits (common.cuh spin_expired) and the fall-back.
struct DeviceLock {
    int fd;
    bool held = false; // fd < 0 (no lock file): nothing to hold, the launch goes ahead
    // Never waits without a bound: LOCK_NB + a deadline.  A holder that is stopped or hung (or a stranger sitting on the file)
    // then costs this solve its persistent path (the caller falls back and counts a give-up), not its progress.
    DeviceLock(int fd_, int wait_ms) : fd(fd_) {
        if (fd < 0) {
            held = true;
            return;
        }
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(wait_ms);
        int nap_us = 20;
        for (;;) {
            if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
                held = true;
                return;
            }
            if (errno != EWOULDBLOCK && errno != EINTR) return; // (a lock that cannot be taken at all: treated as expired)
            if (std::chrono::steady_clock::now() >= deadline) return;
            usleep(nap_us);
            nap_us = std::min(nap_us * 2, 1000);
        }
    }
    ~DeviceLock() {
        if (fd >= 0 && held) (void)flock(fd, LOCK_UN);
    }
    DeviceLock(const DeviceLock &) = delete;
    DeviceLock &operator=(const DeviceLock &) = delete;
};
// Where the lock files live: YALPS_HIP_LOCK_DIR if set, else $XDG_RUNTIME_DIR, else /tmp/yalps_hip_<uid> (created 0700).
// The directory must be a real directory of this user that nobody else can write to -- a lock file in a world-writable
// directory could be pre-planted as a link to one of the user's own files.  (Processes of different users therefore do not
// serialise with each other; against those, as against any foreign kernel, a launch has its bounded waits.)
static bool private_dir_ok(const std::string &dir, bool create) {
    if (create && mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return false;
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
int open_device_lock(int device) {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) std::snprintf(bus, sizeof bus, "dev%d", device);
    for (char *p = bus; *p; p++)
        if (*p == ':' || *p == '/' || *p == '.') *p = '_';
    std::string dir;
    const char *forced = std::getenv("YALPS_HIP_LOCK_DIR");
    const char *xdg = std::getenv("XDG_RUNTIME_DIR");
    if (forced && *forced)
        dir = forced; // (the caller's choice and responsibility; the file checks below still apply)
    else if (xdg && *xdg && private_dir_ok(xdg, false))
        dir = xdg;
    else {
        dir = "/tmp/yalps_hip_" + std::to_string((unsigned long)geteuid());
        if (!private_dir_ok(dir, true)) {
            std::fprintf(stderr, "yalps_hip: %s is not a private directory of this user: persistent launches are not serialised between processes\n", dir.c_str());
            return -1;
        }
    }
    const std::string path = dir + "/yalps_hip_" + bus + ".lock";
    const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0600);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_nlink != 1 || st.st_uid != geteuid()) {
        std::fprintf(stderr, "yalps_hip: cannot use %s as a lock file (%s): persistent launches are not serialised between processes\n",
                     path.c_str(), fd < 0 ? std::strerror(errno) : "not a regular single-link file of this user");
        if (fd >= 0) (void)close(fd);
        return -1;
    }
    return fd;
}
constexpr size_t LDS_DYNAMIC_MAX = 160 * 1024 - 2048;
int allow_big_lds(int device, const void *fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({device, fn})) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYNAMIC_MAX);
    if (e != hipSuccess) return fail(YALPS_E_DEVICE, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    done.insert({device, fn});
    return 0;
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
    bool inplace = true;  // use the persistent in-place kernel beyond that (YALPS_HIP_INPLACE=0: never)
    // A persistent launch whose hand-off gave up (its grid was not co-resident: a foreign kernel held CUs) switches its
    // path off for the next PERSISTENT_RETRY_AFTER solves of this context, not for good; every give-up is counted and
    // reported (stderr once per event, yalps_tableau_info).
    int resident_skip = 0, inplace_skip = 0; // solves left before the path is tried again
    int64_t giveups = 0;
    int lock_fd = -1; // per-device lock file shared by every process that uses this library (persistent launches take turns)
    int lock_wait_ms = 5000; // how long a persistent launch waits for it (YALPS_HIP_LOCK_WAIT_MS); expiry = a give-up
    int64_t lock_giveups = 0;
    int resident_chunk = RESIDENT_CHUNK; // pivots per resident launch (YALPS_HIP_RESIDENT_CHUNK)
    int resident_fault = 0; // test hook: treat the N-th persistent launch on this context as failed (YALPS_HIP_RESIDENT_FAULT=N)
    int64_t persistent_launches = 0;
    int num_cus = 256;
    int max_blocks = 256; // workgroups per launch (one per CU by default)
    // single-workgroup LDS path for small tableaux (YALPS_HIP_SMALL=0: never)
    bool small = true;
    bool small_attr[4] = {false, false, false, false};
    int32_t *small_hist = nullptr; // checkCycles: 2 x small_hist_cap pivot history of small_kernel
    long long small_hist_cap = 0;
    SmallResult *small_res = nullptr; // pinned, written by the kernel over PCIe
    void *small_blob = nullptr;       // pinned staging of the host-array entry point: matrix | pos | var
    size_t small_blob_cap = 0;
};

// Every tableau object, and every re-partition of one, gets an identity that no other ever has: what a cached hipGraph of
// the sharded loop (yalps_comm) is keyed by -- an address can come back after a destroy, device pointers change in set_shard.
static uint64_t next_tableau_generation() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

struct yalps_tableau {
    yalps_ctx *ctx = nullptr;
    uint64_t generation = next_tableau_generation();
    Desc d{};
    int32_t height = 0;
    int cur = 0; // tableau buffer holding the current tableau
    bool generic = false; // no tuned kernel spans this shape: generic_decide_kernel / generic_apply_kernel (in place)
    bool prefer_generic = false; // 8194..16385 columns, unsharded: the any-shape pair is on par with wide_kernel<1024,8>
                                 // on dense tableaux (60 vs 74 us/pivot at 1025x16385, 130 vs 126 at 4097x9001) and skips untouched rows
    int shard_parity = 0;
    RVariant rvar{0, 0, 0, nullptr}; // resident kernel variant, fn == nullptr: tableau does not fit
    int rgen = 1;                    // 2: rvar is a resident2_kernel
    void *rc_sync = nullptr; // flags[2], verdict[2], error word of the persistent kernels (one allocation)
    size_t rc_sync_bytes = 0;
    RVariant svar{0, 0, 0, nullptr}; // stream_kernel variant (persistent, in place)
    RVariant svar_check{0, 0, 0, nullptr}; // the same with hasCycle (options.checkCycles)
    bool sweep = false;                    // svar / svar_check are sweep_kernel variants
    RVariant svar2{0, 0, 0, nullptr};      // stream3_kernel / stream2_kernel variant: delayed row updates (YALPS_HIP_DELAY=0: never)
    RVariant svar2_check{0, 0, 0, nullptr}; // ... with hasCycle (stream3_kernel only)
    bool sattr2_check = false;
    size_t sshmem2 = 0;
    bool sattr2 = false;
    bool last_delayed = true; // what the last in-place solve ran (before the first one: what it would run)
    bool stream3 = false;     // svar2 is a stream3_kernel variant (rows of 8194 .. 16385 columns)
    bool sattr_check = false;
    size_t sshmem = 0;
    bool sattr = false;
    bool occupancy_warned = false;
    int64_t giveups = 0;             // persistent launches of this tableau that gave up waiting (fell back)
    int last_path = 0;               // what the last solve ran: 1 resident, 2 streaming, 4 small, 8 in place (sums: fallbacks)
    int64_t last_launches = 0;       // kernel launches of the last solve that did work (resident: chunks)
    size_t rshmem = 0;
    size_t rx_shmem = 0; // resident kernel with LDS rows: its (fixed) dynamic LDS size
    RVariant rvar_tag{0, 0, 0, nullptr}; // the same variant with tagged candidate rows (fn == nullptr: not for this shape)
    // Control block, ONE device allocation: [flags of both parities | verdict words | granules of the tagged variant |
    // error word (16 B)] [st0 | st1 | cst].  rc_sync / rc_sync_bytes = its first part, zeroed by one memset before every
    // persistent launch; [error word .. st1] comes back in one copy after it; [st0 .. cst] goes up in one copy when a solve
    // starts.  (A fixed ~50 us per device solve was 10 stream operations; this and the one-block basis make it 6.)
    void *ctl_block = nullptr;
    char *host_ctl = nullptr;    // pinned: [st0 | st1 | cst] to send, then [error word | st0 | st1] as received
    int32_t *perm_block = nullptr; // pos[] and var[] in one allocation (pos at 0, var at perm_cap): one copy backs both up
    int32_t perm_cap = 0;
    int32_t *perm_backup = nullptr; // basis before the resident launch in flight (restored if it fails)
    int32_t perm_len = 0; // entries of pos / var (width + GLOBAL height)
    Variant var{};
    int wT_inplace = 0;
    KernelFn dfn = nullptr;         // row shard with delayed row updates: dshard_kernel<512, dJ, nt> (d.dpend / dcolv / dnqv / dlav / dstate)
    int dJ = 0, dnt = 0, dpanel = 0;
    const void *xsweep_fn = nullptr; // dshard_sweep_kernel: the shard's sweep as a launch of its own (dsweep_kernel.cuh)
    int xsweep_grid = 0, shard_pend = 0; // its workgroups (panels x row blocks); pivots the host knows to be pending
    size_t xsweep_lds = 0;
    size_t dshmem = 0;
    void *dsh_block = nullptr;      // ... its arrays, one allocation
    int32_t *cyc_block = nullptr;   // row shard: shard_cycle_kernel's verdict words (Desc::cyc_verdict)
    bool shard_check = false;       // ... the running sharded solve has checkCycles on (yalps_shard_begin)
    KernelFn wfn_inplace = nullptr; // row shard: wide_kernel<.., true, nt> for the MODE_SHARD launches (in place; d.obj holds the objective replicas)
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
    void *cells = nullptr; // staging of yalps_tableau_assemble: row[] col[] val[] of cells_cap entries
    std::vector<char> cut_stage; // host side of yalps_tableau_apply_cuts' one packed upload
    void *pin_out = nullptr;     // pinned staging of yalps_tableau_download_solution
    size_t pin_out_bytes = 0;
    // Internal callers that will want column 0 and the permutations right after the solve (branch-and-cut nodes) ask for
    // them here: the persistent path then enqueues those copies behind every launch, in front of its one wait, and fills
    // the arrays when that launch turns out to be the last (fetch_done) -- no second round trip for the download.
    double *fetch_col0 = nullptr;
    int32_t *fetch_pos = nullptr, *fetch_var = nullptr;
    bool fetch_done = false;
    int64_t cells_cap = 0;
    // branch-and-cut nodes built from a root next door (node_fused_solve)
    void *cut_pin = nullptr; // pinned staging of a node's state block and cuts, read by node_prepare_kernel over PCIe
    size_t cut_pin_bytes = 0;
    bool node_occupancy_ok = false;
    int64_t node_fused_runs = 0;
};

namespace {

void launch_one(yalps_tableau *t, int parity, int mode, int force, const double *gather = nullptr) {
    const int grid = mode == MODE_DECIDE ? 1 : t->nb;
    if (t->wfn && mode != MODE_DECIDE)
        t->wfn<<<dim3(grid), dim3(t->var.T), t->wshmem, t->ctx->stream>>>(t->d, parity, mode, force, gather);
    else
        t->var.fn<<<dim3(grid), dim3(t->var.T), 0, t->ctx->stream>>>(t->d, parity, mode, force, gather);
}

// the elimination step of a row shard (MODE_SHARD): in place where the shard has the kernel for it
void launch_shard(yalps_tableau *t, const double *gathered) {
    const int force = t->ctx->nt_stores ? 64 : 0;
    if (t->shard_check) // checkCycles: the detector's verdict on the pivot this step is about to decide (shard_kernels.cuh)
        shard_cycle_kernel<<<dim3(1), dim3(1024), 0, t->ctx->stream>>>(t->d, t->shard_parity, gathered, (t->dfn || t->wfn_inplace) ? 1 : 0);
    if (t->dfn)
        t->dfn<<<dim3(t->nb), dim3(512), t->dshmem, t->ctx->stream>>>(t->d, t->shard_parity, MODE_SHARD, force, gathered);
    else if (t->wfn_inplace)
        t->wfn_inplace<<<dim3(t->nb), dim3(t->wT_inplace), t->wshmem, t->ctx->stream>>>(t->d, t->shard_parity, MODE_SHARD, force, gathered);
    else
        launch_one(t, t->shard_parity, MODE_SHARD, force, gathered);
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

int init_state(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles, bool wait = true) {
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
    YConst &hc = *reinterpret_cast<YConst *>(t->host_state + 4); // pinned, like the state slots
    std::memset(&hc, 0, sizeof hc);
    hc.height = t->height;
    hc.check_cycles = checkCycles ? 1 : 0;
    hc.hist_cap = t->hist_cap;
    hc.hist_leaving = t->hist[0];
    hc.hist_entering = t->hist[1];
    hc.precision = precision;
    hc.max_pivots = maxPivots;
    YState *hs = &t->host_state[0];
    std::memset(hs, 0, sizeof(YState));
    hs->status = RUNNING;
    hs->phase = 1;
    hs->bootstrap = 1;
    hs->mbuf = t->cur;
    hs->result = NAN;
    if (t->ctl_block) { // [st0 | st1 | cst] are adjacent on the device: one copy (st1 starts out zeroed)
        char *stage = t->host_ctl;
        std::memcpy(stage, hs, sizeof(YState));
        std::memset(stage + sizeof(YState), 0, sizeof(YState));
        std::memcpy(stage + 2 * sizeof(YState), &hc, sizeof(YConst));
        HIP_TRY(hipMemcpyAsync(t->d.st, stage, 2 * sizeof(YState) + sizeof(YConst), hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(t->d.st, hs, sizeof(YState), hipMemcpyHostToDevice, s));
    }
    // (callers that go on to touch slot 0 from the host, or to copy on another stream, wait here; the solve
    // driver does not: everything it enqueues is ordered behind these two copies on the same stream)
    if (wait) HIP_TRY(hipStreamSynchronize(s));
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

// dense-LP(M,N,seed), SURVEY.md 8(d).  tests/helpers/util.ts:20-41 of the reference: the seed is a double,
// += 0x9e3779b9 unwrapped (beyond 2^53 the rounding of that sum is part of the definition, so the stream is walked
// draw by draw even where rows are skipped); ToInt32(seed) = the integer value modulo 2^32.
namespace {
struct DenseLpStream {
    double seed;
    inline uint32_t step() { // the seed's low 32 bits after the increment
        seed += 2654435769.0;
        // (an integer-valued double below 2^63 converts exactly; fmod -- 50 x slower -- for anything else)
        if (seed >= 0.0 && seed < 9223372036854775808.0 && seed == (double)(uint64_t)seed) return (uint32_t)(uint64_t)seed;
        return (uint32_t)(uint64_t)fmod(seed, 4294967296.0);
    }
    inline double next() {
        uint32_t x = step();
        x ^= x >> 16;
        x *= 0x21f0aaadu;
        x ^= x >> 15;
        x *= 0xd35a2d97u;
        x ^= x >> 15;
        return (double)x / 4294967296.0;
    }
};
} // namespace

void yalps_dense_lp_rows_f64(int32_t M, int32_t N, double seed, int32_t row_begin, int32_t row_end, double *rows) {
    // rows [row_begin, row_end) of the (M+1) x (N+1) tableau (row 0 = objective row), row-major into `rows`; the draws of
    // the rows before row_begin are walked (the seed only: a few cycles each), not hashed
    const int32_t w = N + 1, h = M + 1;
    DenseLpStream g{seed};
    if (row_end > h) row_end = h;
    for (int32_t r = 0; r < row_end; r++) {
        const bool keep = r >= row_begin;
        double *mr = keep ? rows + (size_t)(r - row_begin) * w : nullptr;
        if (r == 0) {
            if (keep) mr[0] = 0.0;
        } else if (keep) {
            mr[0] = (double)N * 0.25 * (1.0 + g.next());
        } else {
            g.seed += 2654435769.0;
        }
        if (keep)
            for (int32_t j = 1; j < w; j++) mr[j] = g.next();
        else
            for (int32_t j = 1; j < w; j++) g.seed += 2654435769.0;
    }
}

void yalps_dense_lp_f64(int32_t M, int32_t N, double seed, double *matrix) {
    yalps_dense_lp_rows_f64(M, N, seed, 0, M + 1, matrix);
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
    c->inplace = env_int("YALPS_HIP_INPLACE", 1) != 0;
    c->resident_chunk = env_int("YALPS_HIP_RESIDENT_CHUNK", RESIDENT_CHUNK);
    if (c->resident_chunk < 1) c->resident_chunk = 1;
    c->resident_fault = env_int("YALPS_HIP_RESIDENT_FAULT", 0);
    c->small = env_int("YALPS_HIP_SMALL", 1) != 0;
    c->small_hist_cap = env_int("YALPS_HIP_SMALL_HIST", 16384); // first size of the checkCycles history (test hook)
    if (c->small_hist_cap < 1) c->small_hist_cap = 1;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->small_res), sizeof(SmallResult), hipHostMallocDefault));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; // measured 1-2 % faster in the pivot loop at 2049^2 and 4097^2
    c->lock_fd = env_int("YALPS_HIP_LOCK", 1) ? open_device_lock(device) : -1;
    c->lock_wait_ms = std::max(0, env_int("YALPS_HIP_LOCK_WAIT_MS", 5000));
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
    if (c->small_res) (void)hipHostFree(c->small_res);
    if (c->small_blob) (void)hipHostFree(c->small_blob);
    if (c->small_hist) (void)hipFree(c->small_hist);
    if (c->lock_fd >= 0) (void)close(c->lock_fd);
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
    HIP_TRY(hipSetDevice(ctx->device));
    yalps_tableau *t = new yalps_tableau();
    t->ctx = ctx;
    *out = t; // reachable from now on: the wrapper frees a half-built object on failure
    if (J > 8) { // rows wider than 16385 columns: the any-shape pair of generic_kernels.cuh (run-time loops only)
        t->generic = true;
        T = 1024;
        J = 8;
    }
    if (env_int("YALPS_HIP_GENERIC", 0)) t->generic = true; // (test hook: any shape through the any-shape pair)
    t->prefer_generic = J == 8 && !env_int("YALPS_HIP_WIDE8", 0);
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
    // (pivot_kernel's <1024,2,*> variants spill at the 128-VGPR cap: 47 us/pivot on a 1477x2388 tableau
    // against 27 us for wide_kernel)
    if (J >= 4 || (T == 1024 && J >= 2) || rows_per_block > pick->R || env_int("YALPS_HIP_WIDE", 0)) {
        for (const WVariant &v : kWide)
            if (v.T == T && v.J == J) t->wfn = v.fn;
        t->wshmem = sizeof(double) * ((size_t)((n + 15) / 16 * 16 < 16 ? 16 : (n + 15) / 16 * 16) + 4 * (size_t)rows_per_block); // (+ 2 per row: in-place shards)
        if (t->wshmem > 150 * 1024) t->generic = true; // (pivot row + per-row scalars exceed LDS)
        if (!t->generic && t->wshmem > 48 * 1024)
            if (int rc = allow_big_lds(ctx->device, reinterpret_cast<const void *>(t->wfn))) return rc;
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
    for (int k = 0; k < (t->generic ? 1 : 2); k++) { // (the any-shape pair works in place: one buffer)
        HIP_TRY(hipMalloc(&d.mat[k], mat_bytes));
        HIP_TRY(hipMemsetAsync(d.mat[k], 0, mat_bytes, s));
        HIP_TRY(hipMalloc(&d.rhs[k], sizeof(double) * (size_t)hcap));
    }
    if (t->generic || t->prefer_generic) {
        HIP_TRY(hipMalloc(&d.gen_prow, sizeof(double) * (size_t)d.pitch));
        HIP_TRY(hipMalloc(&d.gen_scal, sizeof(double) * 4));
    }
    t->perm_cap = (width + hcap + 3) / 4 * 4;
    HIP_TRY(hipMalloc(&t->perm_block, sizeof(int32_t) * 2 * (size_t)t->perm_cap));
    d.pos = t->perm_block;
    d.var = t->perm_block + t->perm_cap;
    for (int k = 0; k < 2; k++) {
        HIP_TRY(hipMalloc(&d.part_ratio[k], sizeof(Part) * MAX_BLOCKS));
        HIP_TRY(hipMalloc(&d.part_rhs[k], sizeof(Part) * MAX_BLOCKS));
    }
    // resident (on-chip) solver: needs every workgroup co-resident (one per CU) and the rows of a
    // workgroup in registers
    d.perm_len = t->perm_len;
    if (t->nb <= ctx->num_cus && !t->generic) {
        // (16 waves per CU were tried for 2049^2: <1024,1,9> spills at the 128-VGPR cap and its barriers
        // cost more: 92 K pivots/s against 144 K for <512,2,9>)
        int best = INT_MAX, fT = 0, fJ = 0, fR = 0;
        if (const char *force = std::getenv("YALPS_HIP_RVARIANT")) std::sscanf(force, "%d,%d,%d", &fT, &fJ, &fR);
        for (const RVariant &v : kResident) {
            if (v.T * v.J < units || v.R < rows_per_block) continue;
            if (fT && (v.T != fT || v.J != fJ || v.R != fR)) continue; // experiments: YALPS_HIP_RVARIANT=T,J,R
            const int regs = v.J * v.R * 1024 + v.T; // fewest row registers, then fewest waves
            if (regs < best) {
                best = regs;
                t->rvar = v;
            }
        }
        if (t->rvar.fn && env_int("YALPS_HIP_RESIDENT_GEN", 2) >= 2)
            for (const RVariant &v : kResident2)
                if (v.T == t->rvar.T && v.J == t->rvar.J && v.R == t->rvar.R) {
                    t->rvar = v;
                    t->rgen = 2;
                }
        if (!t->rvar.fn && env_int("YALPS_HIP_LDS_ROWS", 1)) { // a little too tall: park the rows that are missing in LDS
            const int xl_ofs = (2 * (width + hcap) + 3) / 4 * 4;
            int best_extra = INT_MAX;
            for (const RVariant &v : kResidentLds) {
                const int extra = rows_per_block - v.R;
                if (v.T * v.J < units || extra < 1 || extra > XROWS) continue;
                if (fT && (v.T != fT || v.J != fJ || v.R != fR)) continue;
                const size_t bytes = sizeof(int32_t) * (size_t)xl_ofs + sizeof(double) * (size_t)extra * d.pitch;
                if (bytes > LDS_DYNAMIC_MAX || extra >= best_extra) continue;
                best_extra = extra;
                t->rvar = v;
                t->rx_shmem = bytes;
                d.extra = extra;
                d.xl_ofs = xl_ofs;
            }
        }
    }
    // persistent in-place kernel for what does not fit on chip: lanes x units span the row, the normalised
    // pivot row + my rows' scalars fit in LDS
    if (t->nb <= ctx->num_cus && !t->generic) { // (also where a resident variant exists: the fallback order is resident, in place, launches)
        for (const RVariant &v : kStream)
            if (v.T == T && v.J == J) t->svar = v;
        for (const RVariant &v : kStreamCheck)
            if (v.T == T && v.J == J) t->svar_check = v;
        t->sshmem = sizeof(double) * ((size_t)d.pitch + 3 * (size_t)rows_per_block) + sizeof(int32_t) * (size_t)rows_per_block;
        if (t->sshmem > 150 * 1024) t->svar.fn = t->svar_check.fn = nullptr;
        // what streams from HBM anyway goes to sweep_kernel: rows of 8194 .. 16385 columns (no stream_kernel spans them), and
        // 4098 .. 8193-column tableaux too big for the Infinity Cache (measured at 8193 x 8193: stream_kernel 5.1 TB/s)
        const size_t tab_bytes = sizeof(double) * (size_t)d.pitch * hcap;
        const int sweep_mode = env_int("YALPS_HIP_SWEEP", 1); // 0: never, 1: by size, 2: wherever a variant exists
        if (sweep_mode && (J == 8 || (J == 4 && (tab_bytes > SWEEP_BEYOND_CACHE || sweep_mode == 2)))) {
            const size_t lds = sizeof(double) * 2 * 512 * (size_t)(2 * J) + (4 * sizeof(double) + sizeof(int32_t)) * (size_t)rows_per_block + 64;
            if (lds <= 150 * 1024) {
                int want_nt = tab_bytes > SWEEP_BEYOND_CACHE ? 1 : 0;
                if (const char *e = std::getenv("YALPS_HIP_SWEEP_NT")) want_nt = std::atoi(e) != 0;
                const int sT = 512, sJ = 2 * J; // (sweep_kernel runs 512 lanes x 8 / 16 units: persistent_sweep.hip)
                for (const RVariant &v : kSweep)
                    if (v.T == sT && v.J == sJ && v.R == want_nt) {
                        t->svar = v;
                        t->sweep = true;
                    }
                if (t->sweep) {
                    t->svar_check.fn = nullptr;
                    for (const RVariant &v : kSweepCheck)
                        if (v.T == sT && v.J == sJ && v.R == want_nt) t->svar_check = v;
                    t->sshmem = lds;
                    d.sw_nt = want_nt;
                }
            }
        }
    }
    // Delayed row updates: where an in-place kernel applies (and no checkCycles), several pivots per sweep.
    //   stream3_kernel<512, J> (objective replica in LDS, the pending pivot rows in a global scratch shared by all workgroups,
    //   up to 8 of them): rows of up to 16385 columns with at least 4 rows per workgroup;
    //   stream2_kernel (pending rows in LDS, up to 4): fewer rows per workgroup, and where YALPS_HIP_DELAY_KERNEL=2 asks for it.
    // Measured against each other on one box, us per pivot, stream3 / stream2: 4097^2 20.2 / 26.2, 6001^2 33.6 / 50.0,
    // 8193 x 4097 30.0 / 42.8, 8193^2 48.3 / 105, 12001 x 1501 21.8 / 26.4, 11001 x 901 17.2 / 24.0, 16385^2 159 / --;
    // 1025 x 16385: stream3 34.3, sweep_kernel 50.3.
    if (t->svar.fn && env_int("YALPS_HIP_DELAY", 1)) {
        const int units = d.pitch / 2;
        const size_t tab_bytes2 = sizeof(double) * (size_t)d.pitch * hcap;
        int want_nt2 = tab_bytes2 > SWEEP_BEYOND_CACHE ? 1 : 0;
        if (const char *e = std::getenv("YALPS_HIP_DELAY_NT")) want_nt2 = std::atoi(e) != 0;
        int sJ = 0;
        for (int cand : {1, 2, 4, 6, 8, 16})
            if (!sJ && 512 * cand >= units) sJ = cand;
        const int want_kernel = env_int("YALPS_HIP_DELAY_KERNEL", 3);
        if (want_kernel == 3 && sJ && t->nb <= 256 && rows_per_block >= env_int("YALPS_HIP_DELAY_MIN_ROWS", 4)) { // (256: its table of the workgroups' XCDs)
            // depth: a pivot's head grows with the pivots pending (the candidate row gets them all applied before it is
            // published), the sweep shrinks: measured best 8 at 65 and 33 rows per workgroup, 6-8 at 17, 4 at 5-9
            // (with the two-step exchange of the 8- and 16-unit forms only the winner's row gets them applied: 2049 x 16385 at
            // depth 4 / 6 / 8: 48.1 / 46.2 / 44.8 us per pivot, 4097 x 8193 40.6 / 36.7 / 35.1, 4097 x 16385 73.8 / 64.5 / 60.5,
            // 1025 x 16385 31.6 / 31.4 / 32.3)
            // (round 3: the sweep stages the pending rows in LDS one 1024-column panel at a time -- panel_flush.cuh -- and the objective
            // replica moved to registers: up to STREAM3_MAXD = 16 pending pivots, the deepest form whose scalars + panel fit in LDS)
            // The panels pay from STREAM3_PANEL_MIN_ROWS rows per workgroup on (YALPS_HIP_STREAM3_PANEL=0|1 forces); below, the sweep reads
            // the pending rows straight from the XCD's scratch and the depths are round 2's (shape sweep, us per pivot with panels
            // / straight from L2, same build: 4 rows per workgroup -- 1025 x 8001 23.6 / 19.1, 1025 x 16385 38.3 / 28.5; 8 rows -- 2049 x 16385
            // 40.2 / 38.1; 17 rows -- 4097 x 16385 48.9 / 54.9, 4097^2 19.7 / 20.4; 20 rows -- 5001^2 24.8 / 28.2; beyond: panels only,
            // 6001^2 28.1, 8193^2 37.5, 16385^2 87.8 -- round 2: 33.4, 45.5, 141).
            const bool panel3 = env_int("YALPS_HIP_STREAM3_PANEL", rows_per_block >= STREAM3_PANEL_MIN_ROWS ? 1 : 0) != 0;
            const int depth_default = panel3 ? (sJ == 16 ? STREAM3_DEPTH_J16 : STREAM3_DEFAULT_DEPTH_WIDE)
                                             : sJ >= 8 ? (rows_per_block >= 8 ? 12 : 6) : std::min(8, std::max(4, (rows_per_block + 1) / 3)); // (2049 x 16385 from L2 at depth 8 / 12 / 16: 35.8 / 34.8 / 34.9 us per pivot)
            int depth3 = std::min(STREAM3_MAXD, std::max(2, env_int("YALPS_HIP_DELAY_DEPTH", depth_default)));
            auto lds3_of = [&](int dep) {
                return sizeof(double) * (2 * (size_t)dep + 2) * (size_t)rows_per_block + 3 * sizeof(int32_t) * (((size_t)rows_per_block + 3) / 4 * 4) +
                       (panel3 ? sizeof(double) * (size_t)dep * 2 * stream3_panel_units(sJ) : 0);
            };
            const size_t lds3_cap = panel3 ? LDS_DYNAMIC_MAX : 150 * 1024; // (the panels take what the CU has: 160 KB less the kernel's static arrays)
            while (depth3 > 2 && lds3_of(depth3) > lds3_cap) depth3--;
            const size_t lds3 = lds3_of(depth3);
            const int want_r = want_nt2 | (panel3 ? 2 : 0);
            if (lds3 <= lds3_cap)
                for (const RVariant &v : kStream3)
                    if (v.T == 512 && v.J == sJ && v.R == want_r) t->svar2 = v;
            if (t->svar2.fn)
                for (const RVariant &v : kStream3Check) // (checkCycles with rows of 16 units per lane: the form that sweeps straight from L2 -- with the panels it does not fit the registers; same LDS layout, the panel area unused)
                    if (v.T == 512 && v.J == sJ && v.R == (sJ == 16 ? want_nt2 : want_r)) t->svar2_check = v;
            if (t->svar2.fn) {
                t->sshmem2 = lds3;
                d.delay_depth = depth3;
                HIP_TRY(hipMalloc(&d.pend, sizeof(double) * 8 * 2 * (size_t)depth3 * d.pitch)); // (per XCD: two sets of `depth` rows, shared by its workgroups)
                if (sJ >= 6) HIP_TRY(hipMalloc(&d.ob_park, sizeof(double) * (size_t)t->nb * d.pitch)); // (rows of 6+ units per lane: the objective replicas during a sweep)
                t->stream3 = true;
            }
        }
        // (with two or three rows per workgroup there is nothing to save: 257 x 8193 16.6 us delayed against 14.5)
        if (!t->svar2.fn && T * J <= 4096 && rows_per_block >= env_int("YALPS_HIP_DELAY_MIN_ROWS", 4)) { // stream2_kernel: as many pending pivots as LDS holds pivot rows (+ two scalars per row of mine) for, at most 4
            const size_t per_pivot = sizeof(double) * ((size_t)d.pitch + 2 * (size_t)rows_per_block);
            const size_t fixed = sizeof(double) * 2 * (size_t)rows_per_block + sizeof(int32_t) * (size_t)rows_per_block;
            int depth = (int)std::min<size_t>(4, (150 * 1024 - fixed) / per_pivot);
            depth = std::min(depth, std::max(1, env_int("YALPS_HIP_DELAY_DEPTH", 4)));
            if (depth >= 2) {
                for (int nt = want_nt2; nt >= 0 && !t->svar2.fn; nt--) // (the plain form where no non-temporal one is built)
                    for (const RVariant &v : kStream2)
                        if (v.T * v.J == T * J && v.R == nt) t->svar2 = v; // (same row span; its own lane count)
                t->sshmem2 = fixed + (size_t)depth * per_pivot;
                d.delay_depth = depth;
            }
        }
    }
    if (t->rvar.fn || t->svar.fn) {
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipMalloc(&d.rc_rows[k], sizeof(double) * (size_t)t->nb * d.pitch));
            HIP_TRY(hipMalloc(&d.rc_key[k], sizeof(double) * ((size_t)t->nb + 8)));
        }
    }
    {
        static_assert(sizeof(YState) % 8 == 0 && sizeof(YConst) % 8 == 0, "control block layout");
        const bool persistent = t->rvar.fn || t->svar.fn;
        const size_t nflag = persistent ? 2 * (size_t)t->nb : 0;
        // narrow rows: the tagged form of the same variant, if built (YALPS_HIP_TAG=0 switches it off)
        if (t->rvar.fn && !d.extra && env_int("YALPS_HIP_TAG", 1))
            for (const RVariant &v : kResidentTag)
                if (v.T == t->rvar.T && v.J == t->rvar.J && v.R == t->rvar.R) t->rvar_tag = v;
        const size_t tag_row = 2 * (size_t)d.pitch + 2;
        const size_t tag_bytes = t->rvar_tag.fn ? sizeof(double) * 2 * (size_t)t->nb * tag_row : 0;
        const size_t sweep_bytes = t->sweep ? ((size_t)yalps_sweep_sync_bytes() + 15) / 16 * 16 + 32 * 2 * (size_t)t->nb : 0;
        // stream3_kernel's two-step exchange: [nb] XCD ids (int32, padded to 16 B) | [2][nb] slice flags | [2][nb][HP_SCAL] candidate scalars
        const size_t hp_xcc_bytes = t->stream3 ? (sizeof(int32_t) * (size_t)t->nb + 15) / 16 * 16 : 0;
        const size_t hp_flag_bytes = t->stream3 ? sizeof(unsigned long long) * 2 * (size_t)t->nb : 0;
        const size_t hp_scal_bytes = t->stream3 ? sizeof(double) * 2 * (size_t)t->nb * HP_SCAL : 0;
        const size_t hp_bytes = hp_xcc_bytes + hp_flag_bytes + hp_scal_bytes;
        t->rc_sync_bytes = sizeof(unsigned long long) * (2 * nflag + 2) + tag_bytes + sweep_bytes + hp_bytes + 32 + 16; // (a multiple of 16; 32: rc_rowflag)
        HIP_TRY(hipMalloc(&t->ctl_block, t->rc_sync_bytes + 2 * sizeof(YState) + sizeof(YConst)));
        HIP_TRY(hipMemsetAsync(t->ctl_block, 0, t->rc_sync_bytes + 2 * sizeof(YState) + sizeof(YConst), s));
        t->rc_sync = t->ctl_block;
        unsigned long long *base = static_cast<unsigned long long *>(t->ctl_block);
        d.rc_flag[0] = base;
        d.rc_flag[1] = base + nflag;
        d.rc_verdict = base + 2 * nflag;
        if (t->rvar_tag.fn) {
            d.rc_tag[0] = reinterpret_cast<double *>(base + 2 * nflag + 2);
            d.rc_tag[1] = d.rc_tag[0] + (size_t)t->nb * tag_row;
        }
        if (t->sweep) {
            char *sw = reinterpret_cast<char *>(base + 2 * nflag + 2) + tag_bytes;
            d.sw_sync = reinterpret_cast<unsigned long long *>(sw);
            d.sw_recs = reinterpret_cast<unsigned long long *>(sw + ((size_t)yalps_sweep_sync_bytes() + 15) / 16 * 16);
        }
        if (t->stream3) {
            char *hp = reinterpret_cast<char *>(base + 2 * nflag + 2) + tag_bytes + sweep_bytes;
            d.hp_xcc = reinterpret_cast<int32_t *>(hp);
            d.hp_flag = reinterpret_cast<unsigned long long *>(hp + hp_xcc_bytes);
            d.hp_scal = reinterpret_cast<double *>(hp + hp_xcc_bytes + hp_flag_bytes);
        }
        char *tail = static_cast<char *>(t->ctl_block) + t->rc_sync_bytes;
        d.rc_err = reinterpret_cast<int32_t *>(tail - 16);
        d.rc_rowflag = reinterpret_cast<unsigned long long *>(tail - 48);
        d.st = reinterpret_cast<YState *>(tail);
        d.cst = reinterpret_cast<YConst *>(tail + 2 * sizeof(YState));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&t->host_ctl), 16 + 2 * sizeof(YState) + sizeof(YConst), hipHostMallocDefault));
    }
    HIP_TRY(hipHostMalloc(&t->host_state, sizeof(YState) * 4 + sizeof(YConst), hipHostMallocDefault)); // 4 slots + YConst
#ifdef YALPS_STAMPS
    HIP_TRY(hipMalloc(&d.dbg, sizeof(unsigned long long) * STAMP_WORDS * MAX_BLOCKS));
    HIP_TRY(hipMemsetAsync(d.dbg, 0, sizeof(unsigned long long) * STAMP_WORDS * MAX_BLOCKS, s));
#endif
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
    if (t->cut_pin) (void)hipHostFree(t->cut_pin);
    Desc &d = t->d;
    // (pos / var and st / cst / the hand-off words are parts of perm_block and ctl_block where those exist: ordinary tableaux)
    void *bufs[] = {d.mat[0], d.mat[1], d.rhs[0], d.rhs[1], t->perm_block ? nullptr : d.pos, t->perm_block ? nullptr : d.var, t->perm_block,
                    t->ctl_block ? nullptr : d.st, t->ctl_block ? nullptr : d.cst, t->ctl_block, d.rc_rows[0], d.rc_rows[1], t->perm_backup,
                    d.rc_key[0], d.rc_key[1], d.gen_prow, d.gen_scal, d.part_ratio[0], d.part_ratio[1], d.part_rhs[0], d.part_rhs[1],
                    t->hist[0], t->hist[1], t->cells, d.dbg, d.obj[0], d.pend, d.ob_park, t->dsh_block, t->cyc_block};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (t->host_state) (void)hipHostFree(t->host_state);
    if (t->host_ctl) (void)hipHostFree(t->host_ctl);
    if (t->pin_out) (void)hipHostFree(t->pin_out);
    for (auto &e : t->slot_ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

int32_t yalps_tableau_height(const yalps_tableau *t) { return t ? t->height : 0; }

int32_t yalps_tableau_info(const yalps_tableau *t, char *buf, int32_t len) {
    if (!t || !buf || len < 1) return fail(YALPS_E_ARG, "yalps_tableau_info: bad argument");
    char res[96] = "none", inp[64] = "none";
    if (t->rvar.fn)
        std::snprintf(res, sizeof res, "resident%s_kernel<%d,%d,%d%s> chunk=%d lds_rows=%d", t->rgen == 2 && !t->rvar_tag.fn ? "2" : "", t->rvar.T, t->rvar.J, t->rvar.R,
                      t->d.extra ? ",lds" : t->rvar_tag.fn ? ",tag" : "", RESIDENT_CHUNK, t->d.extra);
    if (t->svar2.fn && t->last_delayed)
        std::snprintf(inp, sizeof inp, "stream%d_kernel<%d,%d%s> delay_depth=%d%s", t->stream3 ? 3 : 2, t->svar2.T, t->svar2.J, (t->svar2.R & 1) ? ",nt" : "", t->d.delay_depth,
                      t->stream3 ? ((t->svar2.R & 2) ? " sweep=panels" : " sweep=direct") : "");
    else if (t->svar.fn)
        std::snprintf(inp, sizeof inp, "%s_kernel<%d,%d%s>", t->sweep ? "sweep" : "stream", t->svar.T, t->svar.J, t->sweep && t->d.sw_nt ? ",nt" : "");
    char str[64];
    if (t->dfn)
        std::snprintf(str, sizeof str, "dshard_kernel<512,%d%s%s>,delay_depth:%d", t->dJ, t->dnt ? ",nt" : "", t->dpanel ? ",panel" : "", t->d.delay_depth);
    else if (t->wfn)
        std::snprintf(str, sizeof str, "wide_kernel<%d,%d>", t->var.T, t->var.J);
    else
        std::snprintf(str, sizeof str, "pivot_kernel<%d,%d,%d>", t->var.T, t->var.J, t->var.R);
    std::snprintf(buf, (size_t)len, "streaming=%s workgroups=%d resident=%s inplace=%s giveups=%lld resident_off_for=%d inplace_off_for=%d "
                  "last_path=%s last_resident_launches=%lld node_fused_runs=%lld lock_giveups=%lld decide=pivot_kernel<%d,%d,%d> shard_sweep=%s", str,
                  t->nb, res, inp, (long long)t->giveups, t->ctx->resident ? t->ctx->resident_skip : -1,
                  t->ctx->inplace ? t->ctx->inplace_skip : -1,
                  t->last_path == 1 ? "resident" : t->last_path == 2 ? "streaming" : t->last_path == 3 ? "resident+streaming"
                  : t->last_path == 4 ? "small" : t->last_path == 8 ? "inplace" : t->last_path == 10 ? "inplace+streaming"
                  : t->last_path == 9 ? "resident+inplace" : t->last_path == 11 ? "resident+inplace+streaming"
                  : t->last_path == 16 ? "generic" : "none",
                  (long long)(t->last_path & 9 ? t->last_launches : 0), (long long)t->node_fused_runs, (long long)t->ctx->lock_giveups,
                  t->var.T, t->var.J, t->var.R, !t->dfn ? "none" : t->d.ext_sweep ? "launch" : "inline"); // (decide: the single-workgroup DECIDE launches of checkCycles on the launch-per-pivot path)
    return 0;
}

int32_t yalps_tableau_debug_stamps(yalps_tableau *t, uint64_t *out, int32_t cap_words, int32_t reset) {
#ifdef YALPS_STAMPS
    if (!t || !t->d.dbg || (cap_words > 0 && !out)) return fail(YALPS_E_ARG, "yalps_tableau_debug_stamps: bad argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    const int32_t words = t->nb * STAMP_WORDS < cap_words ? t->nb * STAMP_WORDS : cap_words;
    if (words > 0) HIP_TRY(hipMemcpy(out, t->d.dbg, sizeof(uint64_t) * (size_t)words, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(t->d.dbg, 0, sizeof(unsigned long long) * STAMP_WORDS * MAX_BLOCKS));
    return words;
#else
    (void)t, (void)out, (void)cap_words, (void)reset;
    return fail(YALPS_E_ARG, "yalps_tableau_debug_stamps: this is not the diagnostic build (-DYALPS_STAMPS)");
#endif
}

int32_t yalps_ctx_exchange_floor(yalps_ctx *c, int32_t workgroups, int32_t lanes, int32_t units, int32_t epochs, int32_t variant,
                                 float *us_per_epoch_out) {
    // measurement hook (bench.py: roofline.onchip_floor): `epochs` rounds of the resident kernels' exchange and nothing else
    if (!c || !us_per_epoch_out || workgroups < 1 || workgroups > MAX_BLOCKS || epochs < 1)
        return fail(YALPS_E_ARG, "yalps_ctx_exchange_floor: bad argument");
    using FloorFn = void (*)(double *, unsigned long long *, int32_t *, double *, int, int);
    const FloorFn fn = reinterpret_cast<FloorFn>(const_cast<void *>(yalps_exchange_floor_fn(lanes, units)));
    if (!fn) return fail(YALPS_E_ARG, "yalps_ctx_exchange_floor: no such variant (lanes x units: 512x2, 512x3, 256x1, 256x2)");
    HIP_TRY(hipSetDevice(c->device));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(fn), lanes, 0));
    if (per_cu < 1 || workgroups > c->num_cus * per_cu) return fail(YALPS_E_ARG, "yalps_ctx_exchange_floor: the grid would not be co-resident");
    const size_t row_bytes = sizeof(double) * 2 * (size_t)units * lanes, rows_bytes = 2 * (size_t)workgroups * row_bytes;
    const size_t flag_bytes = 2 * (size_t)workgroups * 16 + 2 * 8 * 16 + (sizeof(int32_t) * (size_t)workgroups + 15) / 16 * 16; // records | XCD records | XCD ids
    const size_t sink_bytes = sizeof(double) * (size_t)workgroups * lanes;
    char *block = nullptr;
    HIP_TRY(hipMalloc(&block, rows_bytes + flag_bytes + 16 + sink_bytes));
    hipStream_t s = c->stream;
    int32_t herr = 0;
    float ms = 0.f;
    hipError_t e = hipMemsetAsync(block, 0, rows_bytes + flag_bytes + 16, s);
    {
        std::lock_guard<std::mutex> one_grid(persistent_mutex(c->device));
        DeviceLock one_grid_of_all_processes(c->lock_fd, c->lock_wait_ms);
        if (e == hipSuccess && !one_grid_of_all_processes.held) {
            (void)hipFree(block);
            return fail(YALPS_E_DEVICE, "yalps_ctx_exchange_floor: the device's lock file is held by another process");
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev0, s);
        if (e == hipSuccess) {
            fn<<<dim3(workgroups), dim3(lanes), 0, s>>>(reinterpret_cast<double *>(block), reinterpret_cast<unsigned long long *>(block + rows_bytes),
                                                        reinterpret_cast<int32_t *>(block + rows_bytes + flag_bytes),
                                                        reinterpret_cast<double *>(block + rows_bytes + flag_bytes + 16), epochs, variant);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev1, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    if (e == hipSuccess) e = hipMemcpy(&herr, block + rows_bytes + flag_bytes, sizeof herr, hipMemcpyDeviceToHost);
    (void)hipFree(block);
    if (e != hipSuccess) return fail(YALPS_E_DEVICE, std::string("yalps_ctx_exchange_floor: ") + hipGetErrorString(e));
    if (herr) return fail(YALPS_E_DEVICE, "yalps_ctx_exchange_floor: the grid gave up waiting (device shared with other work?)");
    *us_per_epoch_out = ms * 1000.f / (float)epochs;
    return 0;
}

int32_t yalps_tableau_padding_check(yalps_tableau *t, int64_t *nonfinite_out, int64_t *nonzero_out) {
    // the pitch - n doubles behind every row of the current buffer (device rows are padded to 128 bytes; the padding is
    // zero after an upload and no kernel has a reason to change it): how many of them are not finite / not zero now
    if (!t || t->height < 1 || !nonfinite_out || !nonzero_out) return fail(YALPS_E_ARG, "yalps_tableau_padding_check: bad argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    const Desc &d = t->d;
    *nonfinite_out = *nonzero_out = 0;
    const int pad = d.pitch - d.n;
    if (pad <= 0) return 0;
    std::vector<double> host((size_t)pad * (size_t)t->height);
    HIP_TRY(hipMemcpy2D(host.data(), sizeof(double) * pad, d.mat[t->cur] + d.n, sizeof(double) * d.pitch, sizeof(double) * pad,
                        (size_t)t->height, hipMemcpyDeviceToHost));
    for (double v : host) {
        if (!std::isfinite(v)) ++*nonfinite_out;
        if (v != 0.0) ++*nonzero_out;
    }
    return 0;
}

int32_t yalps_tableau_upload(yalps_tableau *t, const double *matrix, int32_t height, const int32_t *pos,
                             const int32_t *var) {
    if (!t || !matrix || !pos || !var) return fail(YALPS_E_ARG, "yalps_tableau_upload: NULL argument");
    if (height < 1 || height > t->d.hcap) return fail(YALPS_E_ARG, "yalps_tableau_upload: height exceeds capacity");
    if (t->d.nshards > 1) return fail(YALPS_E_ARG, "yalps_tableau_upload: tableau is sharded; create a new one");
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

int32_t yalps_tableau_assemble(yalps_tableau *t, int32_t height, int64_t nnz, const int32_t *row, const int32_t *col,
                               const double *val) {
    if (!t || nnz < 0 || (nnz > 0 && (!row || !col || !val)))
        return fail(YALPS_E_ARG, "yalps_tableau_assemble: NULL argument");
    if (height < 1 || height > t->d.hcap) return fail(YALPS_E_ARG, "yalps_tableau_assemble: height exceeds capacity");
    if (t->d.nshards > 1) return fail(YALPS_E_ARG, "yalps_tableau_assemble: tableau is sharded; create a new one");
    if (nnz > INT32_MAX) return fail(YALPS_E_ARG, "yalps_tableau_assemble: too many cells");
    const Desc &d = t->d;
    for (int64_t i = 0; i < nnz; i++) { // in range and strictly increasing in (row, col): no cell twice
        if (row[i] < 0 || row[i] >= height || col[i] < 0 || col[i] >= d.w)
            return fail(YALPS_E_ARG, "yalps_tableau_assemble: cell outside the tableau");
        if (i > 0 && (row[i] < row[i - 1] || (row[i] == row[i - 1] && col[i] <= col[i - 1])))
            return fail(YALPS_E_ARG, "yalps_tableau_assemble: cells must be sorted by (row, column) without duplicates");
    }
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    if (nnz > t->cells_cap) {
        if (t->cells) HIP_TRY(hipFree(t->cells));
        t->cells = nullptr;
        t->cells_cap = 0;
        const int64_t cap = nnz + nnz / 2 + 1024;
        HIP_TRY(hipMalloc(&t->cells, (size_t)cap * 16));
        t->cells_cap = cap;
    }
    double *dval = static_cast<double *>(t->cells);
    int32_t *drow = reinterpret_cast<int32_t *>(dval + t->cells_cap), *dcol = drow + t->cells_cap;
    t->cur = 0;
    t->perm_len = d.w + height;
    t->d.perm_len = t->perm_len;
    const size_t total = (size_t)height * d.pitch / 2;
    const int clear_blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)t->ctx->num_cus * 8);
    assemble_clear_kernel<<<dim3(clear_blocks < 1 ? 1 : clear_blocks), dim3(256), 0, s>>>(t->d, height);
    if (nnz > 0) {
        HIP_TRY(hipMemcpyAsync(dval, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(drow, row, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dcol, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, s));
        assemble_scatter_kernel<<<dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s>>>(t->d, (int)nnz, drow, dcol, dval);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    t->height = height;
    return 0;
}

// pinned staging for column 0 + both permutations of the current height
static int ensure_pin_out(yalps_tableau *t) {
    const size_t ncol = sizeof(double) * (size_t)t->height, nperm = sizeof(int32_t) * (size_t)t->perm_len;
    if (ncol + 2 * nperm + sizeof(int32_t) * (size_t)t->perm_cap > t->pin_out_bytes) {
        if (t->pin_out) HIP_TRY(hipHostFree(t->pin_out));
        t->pin_out = nullptr;
        t->pin_out_bytes = 0;
        const size_t cap = sizeof(double) * (size_t)t->d.hcap + 2 * sizeof(int32_t) * ((size_t)t->perm_len + t->d.hcap + t->perm_cap) + 64;
        HIP_TRY(hipHostMalloc(&t->pin_out, cap, hipHostMallocDefault));
        t->pin_out_bytes = cap;
    }
    return 0;
}

int32_t yalps_tableau_download_solution(yalps_tableau *t, double *col0, int32_t *pos, int32_t *var) {
    if (!t || !col0 || !pos || !var) return fail(YALPS_E_ARG, "yalps_tableau_download_solution: NULL argument");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    // through pinned staging: three truly asynchronous copies and one wait (copies into pageable memory block one by one)
    const size_t ncol = sizeof(double) * (size_t)t->height, nperm = sizeof(int32_t) * (size_t)t->perm_len;
    if (int rc = ensure_pin_out(t)) return rc;
    char *stage = static_cast<char *>(t->pin_out);
    HIP_TRY(hipMemcpyAsync(stage, t->d.rhs[t->cur], ncol, hipMemcpyDeviceToHost, s));
    if (t->perm_block) { // pos[] and var[] share one allocation (var at perm_cap): one copy for both
        const size_t span = sizeof(int32_t) * (size_t)t->perm_cap + nperm;
        HIP_TRY(hipMemcpyAsync(stage + ncol, t->perm_block, span, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(col0, stage, ncol);
        std::memcpy(pos, stage + ncol, nperm);
        std::memcpy(var, stage + ncol + sizeof(int32_t) * (size_t)t->perm_cap, nperm);
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(stage + ncol, t->d.pos, nperm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(stage + ncol + nperm, t->d.var, nperm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(col0, stage, ncol);
    std::memcpy(pos, stage + ncol, nperm);
    std::memcpy(var, stage + ncol + nperm, nperm);
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

// (wait = false: the caller goes on to solve dst on the same stream and waits there; the cut arrays are staged by the
// copy call itself -- they are pageable memory -- so nothing of the caller's is read after the return)
static int32_t apply_cuts_impl(yalps_tableau *dst, const yalps_tableau *root, int32_t ncuts, const int32_t *cut_sign,
                               const int32_t *cut_variable, const double *cut_value, bool wait) {
    if (!dst || !root || dst == root || ncuts < 0 || (ncuts > 0 && (!cut_sign || !cut_variable || !cut_value)))
        return fail(YALPS_E_ARG, "yalps_tableau_apply_cuts: bad argument");
    if (root->height < 1 || dst->d.w != root->d.w || dst->ctx != root->ctx || root->d.nshards > 1 || dst->d.nshards > 1 ||
        (int64_t)root->height + ncuts > dst->d.hcap)
        return fail(YALPS_E_ARG, "yalps_tableau_apply_cuts: incompatible tableaux");
    for (int32_t i = 0; i < ncuts; i++)
        if (cut_variable[i] < 0 || cut_variable[i] >= root->d.w + root->height)
            return fail(YALPS_E_ARG, "yalps_tableau_apply_cuts: cut on an unknown variable");
    HIP_TRY(hipSetDevice(dst->ctx->device));
    hipStream_t s = dst->ctx->stream;
    { // rows [0, h0), RHS, both permutations: HBM -> HBM, in stream order (no host wait in between)
        dst->cur = 0;
        HIP_TRY(hipMemcpyAsync(dst->d.mat[0], root->d.mat[root->cur], sizeof(double) * (size_t)root->d.pitch * root->height,
                               hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dst->d.rhs[0], root->d.rhs[root->cur], sizeof(double) * (size_t)root->height,
                               hipMemcpyDeviceToDevice, s));
        const size_t nperm = sizeof(int32_t) * (size_t)(root->d.w + root->height);
        if (dst->perm_block && root->perm_block && dst->perm_cap == root->perm_cap) { // same layout: pos and var in one copy
            HIP_TRY(hipMemcpyAsync(dst->perm_block, root->perm_block, sizeof(int32_t) * (size_t)root->perm_cap + nperm,
                                   hipMemcpyDeviceToDevice, s));
        } else {
            HIP_TRY(hipMemcpyAsync(dst->d.pos, root->d.pos, nperm, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(dst->d.var, root->d.var, nperm, hipMemcpyDeviceToDevice, s));
        }
        dst->height = root->height;
        dst->perm_len = root->perm_len;
        dst->d.perm_len = root->perm_len;
    }
    if (ncuts == 0) {
        if (wait) HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    if (ncuts > dst->cells_cap) {
        if (dst->cells) HIP_TRY(hipFree(dst->cells));
        dst->cells = nullptr;
        dst->cells_cap = 0;
        const int64_t cap = (int64_t)ncuts + 1024;
        HIP_TRY(hipMalloc(&dst->cells, (size_t)cap * 16));
        dst->cells_cap = cap;
    }
    // the three cut arrays in one upload: value[ncuts] | sign[ncuts] | variable[ncuts]
    dst->cut_stage.resize((size_t)ncuts * 16);
    std::memcpy(dst->cut_stage.data(), cut_value, sizeof(double) * (size_t)ncuts);
    std::memcpy(dst->cut_stage.data() + sizeof(double) * (size_t)ncuts, cut_sign, sizeof(int32_t) * (size_t)ncuts);
    std::memcpy(dst->cut_stage.data() + 12 * (size_t)ncuts, cut_variable, sizeof(int32_t) * (size_t)ncuts);
    double *dval = static_cast<double *>(dst->cells);
    int32_t *dsign = reinterpret_cast<int32_t *>(dval + ncuts), *dvar = dsign + ncuts;
    HIP_TRY(hipMemcpyAsync(dval, dst->cut_stage.data(), (size_t)ncuts * 16, hipMemcpyHostToDevice, s));
    const int h0 = root->height;
    apply_cuts_kernel<<<dim3(ncuts), dim3(256), 0, s>>>(dst->d, root->d.mat[root->cur], root->d.rhs[root->cur], root->d.pos, h0,
                                                        ncuts, dsign, dvar, dval);
    HIP_TRY(hipGetLastError());
    if (wait) HIP_TRY(hipStreamSynchronize(s));
    dst->height = h0 + ncuts;
    dst->perm_len = dst->d.w + dst->height;
    dst->d.perm_len = dst->perm_len;
    return 0;
}

int32_t yalps_tableau_apply_cuts(yalps_tableau *dst, const yalps_tableau *root, int32_t ncuts, const int32_t *cut_sign,
                                 const int32_t *cut_variable, const double *cut_value) {
    return apply_cuts_impl(dst, root, ncuts, cut_sign, cut_variable, cut_value, true);
}

// Any-shape fallback (generic_kernels.cuh): batches of DECIDE + APPLY launch pairs, state read back once per batch.
static int32_t solve_generic(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles, double *result_out,
                             int64_t *pivots_out, float *gpu_ms_out) {
    yalps_ctx *c = t->ctx;
    hipStream_t s = c->stream;
    if (t->d.nshards > 1) return fail(YALPS_E_ARG, "row shards of this width are not supported");
    int rc = init_state(t, precision, maxPivots, checkCycles);
    if (rc) return rc;
    const int nb = t->nb, rpw = (t->d.hcap + nb - 1) / nb;
    const size_t shmem = sizeof(double) * (size_t)rpw;
    if (shmem > 150 * 1024) return fail(YALPS_E_ARG, "tableau too tall for this build");
    if (shmem > 48 * 1024)
        if (int rc = allow_big_lds(c->device, reinterpret_cast<const void *>(generic_apply_kernel))) return rc;
    constexpr int PAIRS = 32;
    if (gpu_ms_out) HIP_TRY(hipEventRecord(c->ev0, s));
    t->last_path = 16;
    t->last_launches = 0;
    int64_t hist_have = 0;
    YState fin;
    for (;;) {
        if (checkCycles && hist_have + PAIRS > t->hist_cap) { // room for every pivot a batch can record
            rc = grow_history(t, hist_have + PAIRS, hist_have);
            if (rc) return rc;
            YConst hc;
            HIP_TRY(hipMemcpy(&hc, t->d.cst, sizeof(YConst), hipMemcpyDeviceToHost));
            hc.hist_cap = t->hist_cap;
            hc.hist_leaving = t->hist[0];
            hc.hist_entering = t->hist[1];
            HIP_TRY(hipMemcpy(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice));
        }
        for (int i = 0; i < PAIRS; i++) {
            generic_decide_kernel<<<dim3(1), dim3(1024), 0, s>>>(t->d);
            generic_apply_kernel<<<dim3(nb), dim3(1024), shmem, s>>>(t->d);
        }
        t->last_launches += 2 * PAIRS;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&t->host_state[1], t->d.st, sizeof(YState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        fin = t->host_state[1];
        if (fin.status != RUNNING) break;
        hist_have = fin.hist_len;
    }
    if (gpu_ms_out) {
        HIP_TRY(hipEventRecord(c->ev1, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    }
    t->cur = 0;
    if (result_out) *result_out = fin.result;
    if (pivots_out) *pivots_out = fin.pivots;
    return fin.status;
}

// One launch of small_kernel on the context's stream and the wait for it; the kernel leaves status /
// result / pivot count in pinned host memory.
static int32_t run_small(yalps_ctx *c, SmallDesc sd, int32_t checkCycles, double *result_out, int64_t *pivots_out,
                         float *gpu_ms_out) {
    hipStream_t s = c->stream;
    const size_t shmem = small_lds_bytes(sd.w, sd.h);
    // 16-byte units to sweep per pivot; 1024 lanes from 1024 units on (36x101: 3.8 -> 3.3 us/pivot, 101x61: 5.1 -> 4.0;
    // 64 lanes were slower even at 33x33)
    const int big = (size_t)sd.h * (size_t)(small_pcols(sd.n) / 2) >= 1024 ? 1 : 0;
    const int which = 2 * (checkCycles ? 1 : 0) + big;
    using Fn = void (*)(SmallDesc);
    static const Fn fns[4] = {small_kernel<256, false>, small_kernel<1024, false>, small_kernel<256, true>,
                              small_kernel<1024, true>};
    if (shmem > 48 * 1024 && !c->small_attr[which]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fns[which]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)SMALL_LDS_MAX));
        c->small_attr[which] = true;
    }
    sd.lp = small_lds_pitch(sd.n);
    sd.res = c->small_res;
    if (gpu_ms_out) HIP_TRY(hipEventRecord(c->ev0, s));
    for (;;) {
        if (checkCycles && !c->small_hist) {
            const long long cap = c->small_hist_cap ? c->small_hist_cap : 16384;
            HIP_TRY(hipMalloc(&c->small_hist, sizeof(int32_t) * 2 * (size_t)cap));
            c->small_hist_cap = cap;
        }
        sd.hist_l = c->small_hist;
        sd.hist_e = c->small_hist ? c->small_hist + c->small_hist_cap : nullptr;
        sd.hist_cap = c->small_hist_cap;
        c->small_res->status = YALPS_E_DEVICE;
        fns[which]<<<dim3(1), dim3(big ? 1024 : 256), shmem, s>>>(sd);
        HIP_TRY(hipGetLastError());
        if (gpu_ms_out) HIP_TRY(hipEventRecord(c->ev1, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->small_res->status != WG_HISTORY_FULL) break;
        // a phase ran longer than the history buffer: the kernel left the tableau untouched; grow and rerun
        HIP_TRY(hipFree(c->small_hist));
        c->small_hist = nullptr;
        c->small_hist_cap *= 4;
    }
    if (gpu_ms_out) HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    const SmallResult r = *c->small_res;
    if (r.status < 0) return fail(YALPS_E_DEVICE, "small_kernel did not report a result");
    if (result_out) *result_out = r.result;
    if (pivots_out) *pivots_out = r.pivots;
    return r.status;
}

static bool fits_small(const yalps_ctx *c, int32_t w, int32_t h) {
    return c->small && small_lds_bytes(w, h) <= SMALL_LDS_MAX;
}

int32_t yalps_tableau_solve(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles,
                            double *result_out, int64_t *pivots_out, float *gpu_ms_out) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_tableau_solve: no tableau uploaded");
    yalps_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // (0) the tableau fits in the LDS of one CU: one workgroup, one launch, in place
    if (t->d.nshards == 1 && fits_small(c, t->d.w, t->height)) {
        SmallDesc sd{};
        sd.mat = t->d.mat[t->cur];
        sd.rhs = t->d.rhs[t->cur];
        sd.pitch = t->d.pitch;
        sd.rhs_stride = 1;
        sd.pos = t->d.pos;
        sd.var = t->d.var;
        sd.w = t->d.w;
        sd.n = t->d.n;
        sd.h = t->height;
        sd.precision = precision;
        sd.max_pivots = maxPivots;
        t->last_path = 4;
        t->last_launches = 1;
        return run_small(c, sd, checkCycles, result_out, pivots_out, gpu_ms_out);
    }
    // (a path switched off by a give-up comes back after PERSISTENT_RETRY_AFTER solves)
    const bool resident_on = c->resident && (c->resident_skip == 0 || --c->resident_skip == 0);
    const bool inplace_on = c->inplace && (c->inplace_skip == 0 || --c->inplace_skip == 0);
    // (rows of 8194 .. 16385 columns: sweep_kernel where it applies, else the any-shape pair)
    const bool sweep_ok = t->sweep && t->d.nshards == 1 && inplace_on && (checkCycles ? (t->svar_check.fn || t->svar2_check.fn) : t->svar.fn != nullptr);
    if (t->generic || (t->prefer_generic && t->d.nshards == 1 && !sweep_ok))
        return solve_generic(t, precision, maxPivots, checkCycles, result_out, pivots_out, gpu_ms_out);
    const int which = checkCycles ? 1 : 0;
    int rc = init_state(t, precision, maxPivots, checkCycles, false);
    if (rc) return rc;
    if (gpu_ms_out) HIP_TRY(hipEventRecord(c->ev0, s));
    YState fin;
    std::memset(&fin, 0, sizeof fin);
    bool finished = false;
    t->last_path = 0;
    t->last_launches = 0;

    // (a) persistent kernels, one launch = up to `chunk` pivots: the register-resident kernel when the tableau
    //     fits on chip, else the in-place streaming kernel
    // (fall-back order: resident -> in place -> one launch per pivot; a path that fails is not tried again on this context)
    int64_t hist_have = 0; // checkCycles: pivots recorded in the current phase at the next launch's start
    for (int attempt = 0; attempt < 2 && !finished; attempt++) {
        const bool persistent_ok = t->d.nshards == 1;
        const bool use_resident = persistent_ok && resident_on && c->resident_skip == 0 && t->rvar.fn; // (checkCycles: one more exchange per pivot)
        const bool use_stream = persistent_ok && !use_resident && inplace_on && c->inplace_skip == 0 &&
                                (checkCycles ? (t->svar_check.fn || t->svar2_check.fn) : t->svar.fn != nullptr);
        if (!use_resident && !use_stream) break;
        const bool in_place = use_stream;
        const bool delayed = in_place && (checkCycles ? t->svar2_check.fn : t->svar2.fn); // stream3_kernel / stream2_kernel: several pivots per sweep
        if (in_place) t->last_delayed = delayed;
        const RVariant &pv = delayed ? (checkCycles ? t->svar2_check : t->svar2)
                             : in_place ? (checkCycles ? t->svar_check : t->svar) : t->rvar_tag.fn ? t->rvar_tag : t->rvar;
        bool &sattr = delayed ? (checkCycles ? t->sattr2_check : t->sattr2) : checkCycles ? t->sattr_check : t->sattr;
        const size_t shmem = delayed ? t->sshmem2 : in_place ? t->sshmem : t->rx_shmem ? t->rx_shmem : sizeof(int32_t) * 2 * (size_t)t->perm_len;
        if (in_place ? !sattr : shmem != t->rshmem) {
            if (shmem > 48 * 1024)
                if (int rc2 = allow_big_lds(c->device, reinterpret_cast<const void *>(pv.fn))) return rc2;
            if (in_place)
                sattr = true;
            else
                t->rshmem = shmem;
        }
        // every workgroup of the grid must be resident at once: ask the runtime what fits (registers, LDS, waves) before
        // launching a grid that would wait for workgroups that cannot start
        {
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(pv.fn), pv.T, shmem));
            if (per_cu < 1 || t->nb > c->num_cus * per_cu) {
                if (!t->occupancy_warned)
                    std::fprintf(stderr, "yalps_hip: %s<%d,%d,%d>: %d workgroups do not fit on %d CUs x %d; using the next path\n",
                                 in_place ? "stream_kernel" : "resident_kernel", pv.T, pv.J, pv.R, t->nb, c->num_cus, per_cu);
                t->occupancy_warned = true;
                if (delayed)
                    t->svar2.fn = t->svar2_check.fn = nullptr;
                else if (in_place)
                    t->svar.fn = t->svar_check.fn = nullptr;
                else
                    t->rvar.fn = t->rvar_tag.fn = nullptr;
                continue;
            }
        }
        int chunk = c->resident_chunk;
        if (in_place) { // bound a launch to ~0.25 s: a pivot streams at most the whole tableau (~6 TB/s), never under ~8 us
            const double us = std::max(8.0, 16.0 * (double)t->height * t->d.w / 6e6);
            chunk = (int)std::min<double>(chunk, std::max(64.0, 250000.0 / us));
        }
        int parity = 0;
        int32_t *herr = reinterpret_cast<int32_t *>(&t->host_state[3]); // pinned scratch
        bool lock_expired = false;
        for (;;) {
            HIP_TRY(hipMemsetAsync(t->rc_sync, 0, t->rc_sync_bytes, s)); // (flags, verdicts, tags -- epochs restart at 1 --, error word)
            if (checkCycles) { // room for every pivot this launch can record (no pause inside a persistent launch)
                const int64_t have = hist_have;
                if (have + chunk > t->hist_cap) {
                    rc = grow_history(t, have + chunk, have);
                    if (rc) return rc;
                    YConst hc;
                    HIP_TRY(hipMemcpy(&hc, t->d.cst, sizeof(YConst), hipMemcpyDeviceToHost));
                    hc.hist_cap = t->hist_cap;
                    hc.hist_leaving = t->hist[0];
                    hc.hist_entering = t->hist[1];
                    HIP_TRY(hipMemcpy(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice));
                }
            }
            // the kernel rewrites the basis (and, in place, the tableau): keep the old ones until the launch is known good
            // The resident kernel writes pos / var only where workgroup 0 leaves cleanly, and once it has, every workgroup has
            // everything it needs to finish (all keys of the last epoch are out): a launch that reports a failed hand-off
            // has not touched them.  In place (and for the test hook that declares a good launch failed) keep a copy.
            const bool backup = in_place || c->resident_fault > 0;
            if (backup) {
                if (!t->perm_backup) HIP_TRY(hipMalloc(&t->perm_backup, sizeof(int32_t) * 2 * (size_t)t->perm_cap));
                HIP_TRY(hipMemcpyAsync(t->perm_backup, t->perm_block, sizeof(int32_t) * 2 * (size_t)t->perm_cap, hipMemcpyDeviceToDevice, s));
            }
            if (in_place) {
                const Desc &d = t->d;
                HIP_TRY(hipMemcpyAsync(d.mat[t->cur ^ 1], d.mat[t->cur], sizeof(double) * (size_t)d.pitch * t->height,
                                       hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemcpyAsync(d.rhs[t->cur ^ 1], d.rhs[t->cur], sizeof(double) * (size_t)t->height,
                                       hipMemcpyDeviceToDevice, s));
            }
            {
                // A persistent kernel needs every workgroup of ITS grid on the chip.  Two of them from two contexts
                // (threads) could each get half of the CUs and wait for the rest until their bounded spins give up:
                // within this process, one at a time per device, from the launch to its completion.
                std::lock_guard<std::mutex> one_grid(persistent_mutex(c->device));
                DeviceLock one_grid_of_all_processes(c->lock_fd, c->lock_wait_ms);
                lock_expired = !one_grid_of_all_processes.held;
                if (!lock_expired) { // (else nothing of this launch is enqueued but the copies aside)
                pv.fn<<<dim3(t->nb), dim3(pv.T), shmem, s>>>(t->d, parity, chunk);
                t->last_path |= in_place ? 8 : 1;
                t->last_launches++;
                c->persistent_launches++;
                HIP_TRY(hipGetLastError());
                // [error word | st0 | st1] are adjacent: one copy back
                HIP_TRY(hipMemcpyAsync(t->host_ctl, t->d.rc_err, 16 + 2 * sizeof(YState), hipMemcpyDeviceToHost, s));
                const bool fetch = t->fetch_col0 && t->perm_block;
                const size_t f_ncol = sizeof(double) * (size_t)t->height, f_nperm = sizeof(int32_t) * (size_t)t->perm_len;
                if (fetch) { // (in case this launch is the last one: where it leaves column 0, and the basis)
                    if (int rc2 = ensure_pin_out(t)) return rc2;
                    char *stage = static_cast<char *>(t->pin_out);
                    HIP_TRY(hipMemcpyAsync(stage, t->d.rhs[in_place ? t->cur : t->cur ^ 1], f_ncol, hipMemcpyDeviceToHost, s));
                    HIP_TRY(hipMemcpyAsync(stage + f_ncol, t->perm_block, sizeof(int32_t) * (size_t)t->perm_cap + f_nperm, hipMemcpyDeviceToHost, s));
                }
                HIP_TRY(hipStreamSynchronize(s));
                if (fetch && *reinterpret_cast<int32_t *>(t->host_ctl) == 0 && !(c->resident_fault > 0 && c->persistent_launches == c->resident_fault) &&
                    reinterpret_cast<YState *>(t->host_ctl + 16)[parity ^ 1].status != RUNNING) {
                    const char *stage = static_cast<const char *>(t->pin_out);
                    std::memcpy(t->fetch_col0, stage, f_ncol);
                    std::memcpy(t->fetch_pos, stage + f_ncol, f_nperm);
                    std::memcpy(t->fetch_var, stage + f_ncol + sizeof(int32_t) * (size_t)t->perm_cap, f_nperm);
                    t->fetch_done = true;
                }
                std::memcpy(herr, t->host_ctl, sizeof(int32_t));
                std::memcpy(&t->host_state[1], t->host_ctl + 16 + (size_t)(parity ^ 1) * sizeof(YState), sizeof(YState));
                }
            }
            if (std::getenv("YALPS_HIP_DEBUG")) {
                const YState &hs = t->host_state[1];
                std::fprintf(stderr, "yalps_hip: persistent launch %lld err=%d status=%d phase=%d iter=%g pivots=%lld result=%g mbuf=%d chunk=%d\n",
                             (long long)t->last_launches, *herr, hs.status, hs.phase, hs.iter, (long long)hs.pivots, hs.result, hs.mbuf, chunk);
            }
            if (lock_expired) {
                HIP_TRY(hipStreamSynchronize(s));
                c->lock_giveups++;
                *herr = 0;
            }
            if (lock_expired || *herr || (c->resident_fault > 0 && c->persistent_launches == c->resident_fault)) {
                // a workgroup gave up waiting (grid not co-resident: somebody else's kernel holds CUs): carry on with the
                // next path from the last consistent state (st[parity]; in place: the copy made above), say so, and
                // leave this path alone for the next few solves of this context
                c->giveups++;
                t->giveups++;
                std::fprintf(stderr, "yalps_hip: %s launch %s; "
                                     "falling back for this and the next %d solves (event %lld on this context)\n",
                             in_place ? "stream_kernel" : "resident_kernel",
                             lock_expired ? "did not get the device's lock file in time (another process of this library holds it)"
                                          : "gave up waiting for its grid (device shared with other work?)",
                             PERSISTENT_RETRY_AFTER, (long long)c->giveups);
                if (in_place || lock_expired) // (every persistent path needs the lock)
                    c->inplace_skip = PERSISTENT_RETRY_AFTER + 1;
                if (!in_place || lock_expired)
                    c->resident_skip = PERSISTENT_RETRY_AFTER + 1;
                if (backup)
                    HIP_TRY(hipMemcpyAsync(t->perm_block, t->perm_backup, sizeof(int32_t) * 2 * (size_t)t->perm_cap, hipMemcpyDeviceToDevice, s));
                YState last;
                HIP_TRY(hipMemcpy(&last, t->d.st + parity, sizeof(YState), hipMemcpyDeviceToHost));
                t->cur = in_place ? last.mbuf ^ 1 : last.mbuf;
                if (t->cur != 0) {
                    const Desc &d = t->d;
                    HIP_TRY(hipMemcpyAsync(d.mat[0], d.mat[1], sizeof(double) * (size_t)d.pitch * t->height,
                                           hipMemcpyDeviceToDevice, s));
                    HIP_TRY(hipMemcpyAsync(d.rhs[0], d.rhs[1], sizeof(double) * (size_t)t->height,
                                           hipMemcpyDeviceToDevice, s));
                    t->cur = 0;
                }
                hist_have = last.hist_len;
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
            t->cur = t->host_state[1].mbuf; // (resident launches flip the buffer; in place it stays)
            hist_have = t->host_state[1].hist_len;
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
    if (gpu_ms_out) HIP_TRY(hipEventRecord(c->ev1, s));
    // (the persistent paths have waited for their last launch; the launch-per-pivot loop may still have a
    // batch of no-op launches and a state copy in flight)
    if (gpu_ms_out || (t->last_path & 2)) HIP_TRY(hipStreamSynchronize(s));
    if (gpu_ms_out) HIP_TRY(hipEventElapsedTime(gpu_ms_out, c->ev0, c->ev1));
    t->cur = fin.mbuf;
    if (result_out) *result_out = fin.result;
    if (pivots_out) *pivots_out = fin.pivots;
    return fin.status;
}

// A branch-and-cut node in THREE launches and one wait (src/branchAndCut.ts:22-61 applyCuts + :127 simplex on the node):
// what apply_cuts_impl + yalps_tableau_solve enqueue one by one for a node whose tableau takes the resident kernel --
// three root -> node copies, the cuts, apply_cuts_kernel, the state block, the memset of the hand-off words, the kernel,
// and three copies back: eleven stream operations, every copy a blit kernel of the runtime with ~4.5 us between dependent
// ones (rocprof timeline: ~90 us of device time around a 15 us solve) -- as node_prepare_kernel, the resident kernel and
// node_finish_kernel; cuts, state and results travel through pinned host memory that the kernels read and write directly.
// (The same eleven operations captured as ONE hipGraph per number of cuts ran in exactly the same time, 70.7 vs 71.0 us per
// Monster 2 node: the graph replays the same blit kernels with the same gaps.  YALPS_HIP_NODE_FUSED=0: call by call.)
//   returns 1: done (*status_out, *result_out; column 0 and the basis are in the fetch_* arrays of dst);
//           0: not for this node, or the launch did not finish the solve (give-up, > chunk pivots): the caller runs
//              the node through the ordinary calls, which rebuild it from the root;
//         < 0: error.
static int32_t node_fused_solve(yalps_tableau *dst, const yalps_tableau *root, int32_t ncuts, const int32_t *cut_sign,
                                const int32_t *cut_variable, const double *cut_value, double precision, double maxPivots,
                                int32_t checkCycles, double *result_out, int32_t *status_out) {
    yalps_ctx *c = dst->ctx;
    const bool enabled = env_int("YALPS_HIP_NODE_FUSED", 1) != 0; // (read per call: tests compare both ways in one process)
    if (!enabled || checkCycles || ncuts < 1 || !dst->fetch_col0 || !dst->ctl_block || !dst->perm_block || !root->perm_block ||
        dst->generic || dst->d.nshards != 1 || root->d.nshards != 1 || dst->d.w != root->d.w || dst->ctx != root->ctx ||
        (int64_t)root->height + ncuts > dst->d.hcap || !c->resident || c->resident_skip != 0 || c->resident_fault > 0 ||
        !dst->rvar.fn || fits_small(c, dst->d.w, root->height + ncuts))
        return 0;
    for (int32_t i = 0; i < ncuts; i++)
        if (cut_variable[i] < 0 || cut_variable[i] >= root->d.w + root->height)
            return fail(YALPS_E_ARG, "yalps_tableau_apply_cuts: cut on an unknown variable");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int32_t h0 = root->height, h = h0 + ncuts;
    constexpr size_t STATE_BYTES = 2 * sizeof(YState) + sizeof(YConst);
    static_assert(STATE_BYTES % 16 == 0 && (16 + 2 * sizeof(YState)) % 16 == 0, "moved as 16-byte words");
    if (STATE_BYTES + (size_t)ncuts * 16 > dst->cut_pin_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        if (dst->cut_pin) HIP_TRY(hipHostFree(dst->cut_pin));
        dst->cut_pin = nullptr;
        dst->cut_pin_bytes = 0;
        const size_t cap = STATE_BYTES + ((size_t)ncuts + 1024) * 16;
        HIP_TRY(hipHostMalloc(&dst->cut_pin, cap, hipHostMallocDefault));
        dst->cut_pin_bytes = cap;
    }
    // the tableau object as apply_cuts_impl + init_state leave it
    dst->cur = 0;
    dst->height = h;
    dst->perm_len = dst->d.w + h;
    dst->d.perm_len = dst->perm_len;
    if (int rc = ensure_pin_out(dst)) return rc;
    char *stage = static_cast<char *>(dst->cut_pin); // [st0 | st1 | cst] value[ncuts] | sign[ncuts] | variable[ncuts]
    {
        YConst hc;
        std::memset(&hc, 0, sizeof hc);
        hc.height = h;
        hc.precision = precision;
        hc.max_pivots = maxPivots;
        YState hs;
        std::memset(&hs, 0, sizeof hs);
        hs.status = RUNNING;
        hs.phase = 1;
        hs.bootstrap = 1;
        hs.mbuf = 0;
        hs.result = NAN;
        std::memcpy(stage, &hs, sizeof(YState));
        std::memset(stage + sizeof(YState), 0, sizeof(YState));
        std::memcpy(stage + 2 * sizeof(YState), &hc, sizeof(YConst));
    }
    char *cuts = stage + STATE_BYTES;
    std::memcpy(cuts, cut_value, sizeof(double) * (size_t)ncuts);
    std::memcpy(cuts + sizeof(double) * (size_t)ncuts, cut_sign, sizeof(int32_t) * (size_t)ncuts);
    std::memcpy(cuts + 12 * (size_t)ncuts, cut_variable, sizeof(int32_t) * (size_t)ncuts);
    const RVariant &pv = dst->rvar_tag.fn ? dst->rvar_tag : dst->rvar;
    const size_t shmem = dst->rx_shmem ? dst->rx_shmem : sizeof(int32_t) * 2 * (size_t)dst->perm_len;
    if (shmem != dst->rshmem) {
        if (shmem > 48 * 1024)
            if (int rc2 = allow_big_lds(c->device, reinterpret_cast<const void *>(pv.fn))) return rc2;
        dst->rshmem = shmem;
    }
    if (!dst->node_occupancy_ok) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(pv.fn), pv.T, shmem));
        if (per_cu < 1 || dst->nb > c->num_cus * per_cu) return 0; // (the ordinary path reports it and moves on)
        dst->node_occupancy_ok = true;
    }
    const int prep_blocks = std::max<int>(ncuts, (int)std::min<int64_t>(1024, ((int64_t)root->d.pitch * h0 / 2 + 255) / 256));
    node_prepare_kernel<<<dim3(prep_blocks), dim3(256), 0, s>>>(dst->d, root->d.mat[root->cur], root->d.rhs[root->cur], root->d.pos, root->d.var, h0,
                                                               ncuts, stage, (int)STATE_BYTES, static_cast<unsigned long long *>(dst->rc_sync),
                                                               (long long)(dst->rc_sync_bytes / 8));
    HIP_TRY(hipGetLastError());
    {
        std::lock_guard<std::mutex> one_grid(persistent_mutex(c->device));
        DeviceLock one_grid_of_all_processes(c->lock_fd, c->lock_wait_ms);
        if (!one_grid_of_all_processes.held) { // the ordinary calls rebuild the node and take the next path
            HIP_TRY(hipStreamSynchronize(s));
            c->giveups++;
            c->lock_giveups++;
            c->resident_skip = PERSISTENT_RETRY_AFTER + 1;
            std::fprintf(stderr, "yalps_hip: node solve did not get the device's lock file in time; falling back for this and the next %d solves\n",
                         PERSISTENT_RETRY_AFTER);
            return 0;
        }
        pv.fn<<<dim3(dst->nb), dim3(pv.T), shmem, s>>>(dst->d, 0, c->resident_chunk);
        HIP_TRY(hipGetLastError());
        node_finish_kernel<<<dim3(8), dim3(256), 0, s>>>(dst->d, h, dst->perm_len, dst->perm_cap, (int)(16 + 2 * sizeof(YState)), dst->host_ctl,
                                                         static_cast<char *>(dst->pin_out));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
    }
    dst->last_path = 1;
    dst->last_launches = 1;
    c->persistent_launches++;
    dst->node_fused_runs++;
    const int32_t err = *reinterpret_cast<const int32_t *>(dst->host_ctl);
    const YState fin = reinterpret_cast<const YState *>(dst->host_ctl + 16)[1];
    if (err || fin.status == RUNNING) return 0;
    const size_t f_ncol = sizeof(double) * (size_t)h, f_nperm = sizeof(int32_t) * (size_t)dst->perm_len;
    const char *res = static_cast<const char *>(dst->pin_out);
    std::memcpy(dst->fetch_col0, res, f_ncol);
    std::memcpy(dst->fetch_pos, res + f_ncol, f_nperm);
    std::memcpy(dst->fetch_var, res + f_ncol + sizeof(int32_t) * (size_t)dst->perm_cap, f_nperm);
    dst->fetch_done = true;
    dst->cur = fin.mbuf;
    if (result_out) *result_out = fin.result;
    *status_out = fin.status;
    return 1;
}

int32_t yalps_tableau_node_solve(yalps_tableau *node, const yalps_tableau *root, int32_t ncuts, const int32_t *cut_sign,
                                 const int32_t *cut_variable, const double *cut_value, double precision, double maxPivots,
                                 int32_t checkCycles, double *result_out, double *col0_out, int32_t *pos_out, int32_t *var_out) {
    if (!node || !root || !col0_out || !pos_out || !var_out) return fail(YALPS_E_ARG, "yalps_tableau_node_solve: NULL argument");
    // column 0 and the permutations ride along with the solve's own wait where the persistent path allows it
    node->fetch_col0 = col0_out;
    node->fetch_pos = pos_out;
    node->fetch_var = var_out;
    node->fetch_done = false;
    int32_t st = 0;
    int32_t rc = 0;
    if (node != root && ncuts >= 1 && cut_sign && cut_variable && cut_value)
        rc = node_fused_solve(node, root, ncuts, cut_sign, cut_variable, cut_value, precision, maxPivots, checkCycles, result_out, &st);
    if (rc == 0) {
        node->fetch_done = false;
        rc = apply_cuts_impl(node, root, ncuts, cut_sign, cut_variable, cut_value, false); // (the solve below waits)
        if (rc == 0) st = yalps_tableau_solve(node, precision, maxPivots, checkCycles, result_out, nullptr, nullptr);
    }
    const bool fetched = node->fetch_done;
    node->fetch_col0 = nullptr;
    node->fetch_pos = node->fetch_var = nullptr;
    if (rc < 0) return rc;
    if (st == YALPS_OPTIMAL && !fetched)
        if (int rc2 = yalps_tableau_download_solution(node, col0_out, pos_out, var_out)) return rc2;
    return st;
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
    if (t->generic) return fail(YALPS_E_ARG, "yalps_tableau_pivot: not available for tableaux wider than 16385 columns");
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
    if (t->generic) return fail(YALPS_E_ARG, "yalps_tableau_bench_sweep: not available for tableaux wider than 16385 columns");
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
    if (t->generic) return fail(YALPS_E_ARG, "yalps_tableau_set_shard: row shards wider than 16385 columns are not supported");
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    t->generation = next_tableau_generation(); // (the device arrays below are reallocated: a batch captured before is stale)
    Desc &d = t->d;
    d.nshards = nranks;
    d.shard_rank = rank;
    d.row_base = bounds[rank] - 1;
    for (int k = 0; k <= MAX_SHARDS; k++) d.bounds[k] = k <= nranks ? bounds[k] : INT_MAX;
    // the permutations are global (every rank replays the same basis swaps)
    const size_t n = (size_t)d.w + (size_t)global_height;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipFree(t->perm_block)); // (pos and var share one allocation)
    t->perm_block = nullptr;
    if (t->perm_backup) HIP_TRY(hipFree(t->perm_backup));
    t->perm_backup = nullptr;
    t->perm_cap = (int32_t)((n + 3) / 4 * 4);
    HIP_TRY(hipMalloc(&t->perm_block, sizeof(int32_t) * 2 * (size_t)t->perm_cap));
    d.pos = t->perm_block;
    d.var = t->perm_block + t->perm_cap;
    HIP_TRY(hipMemcpyAsync(d.pos, pos, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d.var, var, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    t->perm_len = (int32_t)n;
    d.perm_len = t->perm_len;
    t->rvar.fn = nullptr; // a shard is driven step by step
    // rows wide enough for wide_kernel are swept in place (YALPS_HIP_SHARD_INPLACE=0: ping-pong as in round 1)
    t->wfn_inplace = nullptr;
    if (t->wfn && env_int("YALPS_HIP_SHARD_INPLACE", 1)) {
        const bool nt = env_int("YALPS_HIP_SHARD_NT", sizeof(double) * (size_t)d.pitch * (size_t)t->height > SWEEP_BEYOND_CACHE ? 1 : 0) != 0;
        for (const WInplace &v : kWideInplace)
            if (v.T == t->var.T && v.J == t->var.J) {
                t->wfn_inplace = v.fn[nt ? 1 : 0];
                t->wT_inplace = v.launchT;
            }
        if (t->wfn_inplace) {
            if (t->wshmem > 48 * 1024)
                if (int rc = allow_big_lds(t->ctx->device, reinterpret_cast<const void *>(t->wfn_inplace))) return rc;
            if (!d.obj[0]) {
                HIP_TRY(hipMalloc(&d.obj[0], sizeof(double) * 2 * (size_t)d.pitch));
                d.obj[1] = d.obj[0] + d.pitch;
            }
        }
    }
    // Delayed row updates (dshard_kernel.cuh): a pivot costs the shard its scalars, the sweep comes once per `depth` pivots.
    // Shards swept in place with at least YALPS_HIP_DELAY_MIN_ROWS rows per workgroup; YALPS_HIP_SHARD_DELAY=0: one sweep per pivot.
    t->dfn = nullptr;
    t->xsweep_fn = nullptr;
    {
        const int rows_per_block = (d.hcap + t->nb - 1) / t->nb, units = d.pitch / 2;
        int dJ = 0;
        for (int cand : {1, 2, 4, 6, 8, 16})
            if (!dJ && 512 * cand >= units) dJ = cand;
        if (t->wfn_inplace && dJ && env_int("YALPS_HIP_SHARD_DELAY", 1) && rows_per_block >= env_int("YALPS_HIP_DELAY_MIN_ROWS", 4)) {
            // (measured, us per pivot at depth 2 / 4 / 6 / 8: 2049 x 16385 73 / 53 / 48 / 46, 8193 x 16385 207 / 125 / 105 / 97:
            // a launch-per-pivot step has a larger fixed part than stream3_kernel's, the deepest form wins everywhere)
            // round 3: the sweep stages the pending rows in LDS one 1024-column panel at a time (panel_flush.cuh) -- up to 16
            // pending pivots; the deepest form whose scalars + panel fit the LDS of a CU
            // the sweep through LDS panels pays from DSHARD_PANEL_MIN_ROWS rows per workgroup on (below: the pending rows straight from L2,
            // at most 8 pending pivots) -- YALPS_HIP_SHARD_PANEL=0|1 forces one or the other
            const bool panel = env_int("YALPS_HIP_SHARD_PANEL", rows_per_block >= DSHARD_PANEL_MIN_ROWS ? 1 : 0) != 0;
            // (measured, one rank, 16385 columns, us per pivot: 2049 rows -- 8 per workgroup -- straight from L2 48 at depth 8, panels 52;
            // 4097 rows 63 / 59.5 at depth 8 / 16 from L2, panels 60; 8193 rows 91 from L2, panels 86 / 74 at depth 8 / 16; 16385 rows panels 102)
            // (round 3, after the panel sweep lost its chains of round trips -- one rank, 16385 columns, us per pivot from L2 / through panels at
            // depth 8, 12, 16: 2049 rows (8 per workgroup) 45.3 43.8 43.3 / 48.3 46.9 46.3; 4097 rows (16) 60.5 57.7 56.6 / 56.0 51.0 49.1;
            // 6001 rows (24) 74.0 70.3 74.5 / 70.7 61.0 58.5)
            const int depth_default = panel ? DSHARD_DEFAULT_DEPTH_PANEL : rows_per_block >= 8 ? 16 : DSHARD_DEFAULT_DEPTH;
            int depth = std::min(DSHARD_MAXD, std::max(2, env_int("YALPS_HIP_DELAY_DEPTH", depth_default)));
            auto lds_of = [&](int dep) {
                return sizeof(double) * (2 * (size_t)dep + 2) * (size_t)rows_per_block + 3 * sizeof(int32_t) * (((size_t)rows_per_block + 3) / 4 * 4) +
                       (panel ? sizeof(double) * (size_t)dep * 2 * DSHARD_PANEL_UNITS : 0);
            };
            while (depth > 2 && lds_of(depth) > 150 * 1024) depth--;
            const bool nt = env_int("YALPS_HIP_SHARD_NT", sizeof(double) * (size_t)d.pitch * (size_t)t->height > SWEEP_BEYOND_CACHE ? 1 : 0) != 0;
            const size_t lds = lds_of(depth);
            if (lds <= 150 * 1024)
                for (const RVariant &v : kDshard)
                    if (v.T == 512 && v.J == dJ && v.R == ((nt ? 1 : 0) | (panel ? 2 : 0))) t->dfn = reinterpret_cast<KernelFn>(v.fn);
            if (t->dfn) {
                if (lds > 48 * 1024)
                    if (int rc = allow_big_lds(t->ctx->device, reinterpret_cast<const void *>(t->dfn))) return rc;
                t->dshmem = lds;
                t->dJ = dJ;
                t->dnt = nt ? 1 : 0;
                t->dpanel = panel ? 1 : 0;
                d.delay_depth = depth;
                // the sweep as a launch of its own, rows mapped to workgroups by (panel, row block): one fill per workgroup, no barrier
                // between its waves afterwards -- whatever the rows per workgroup of the step kernel (YALPS_HIP_SHARD_XSWEEP=0: the step kernel sweeps)
                // Measured, one rank, 16385 columns, us per pivot with the sweep inside the step kernel / as its own launch: 2049 rows (8 per
                // workgroup) 42.9 / 38.7, 4097 (16) 49.6 / 46.6, 8193 (32) 61.2 / 63.0, 16385 (64) 93.3 / 95.8 (the launch sweeps at 3.6-3.9 TB/s
                // whatever the rows; a row-major copy of the coefficients for it -- one cache line per row instead of 32 lines 131 KB apart per pair of
                // rows -- changed nothing: 96.0 / 62.3 / 46.3 / 39.2) -- hence below 24 rows per workgroup.
                t->xsweep_fn = env_int("YALPS_HIP_SHARD_XSWEEP", rows_per_block < DSHARD_XSWEEP_BELOW_ROWS ? 1 : 0) ? yalps_dshard_sweep_fn(nt ? 1 : 0) : nullptr;
                if (t->xsweep_fn) {
                    const int npan = (d.pitch / 2 + DSHARD_PANEL_UNITS - 1) / DSHARD_PANEL_UNITS;
                    t->xsweep_grid = npan * std::max(1, 256 / npan);
                    t->xsweep_lds = sizeof(double) * 2 * DSHARD_PANEL_UNITS * (size_t)depth;
                    if (int rc = allow_big_lds(t->ctx->device, t->xsweep_fn)) return rc;
                }
                if (t->dsh_block) HIP_TRY(hipFree(t->dsh_block));
                t->dsh_block = nullptr;
                const size_t hc = ((size_t)d.hcap + 1) / 2 * 2; // (16-byte parts)
                const size_t doubles = (size_t)depth * d.pitch + 2 * (size_t)depth * hc + hc;
                HIP_TRY(hipMalloc(&t->dsh_block, sizeof(double) * doubles + 2 * sizeof(DelayState)));
                HIP_TRY(hipMemsetAsync(t->dsh_block, 0, sizeof(double) * doubles + 2 * sizeof(DelayState), s));
                d.dpend = static_cast<double *>(t->dsh_block);
                d.dcolv = d.dpend + (size_t)depth * d.pitch;
                d.dnqv = d.dcolv + (size_t)depth * hc;
                d.dlav = d.dnqv + (size_t)depth * hc;
                d.dstate = reinterpret_cast<DelayState *>(d.dlav + hc);
            }
        }
    }
    if (!t->cyc_block) {
        HIP_TRY(hipMalloc(&t->cyc_block, 16));
        HIP_TRY(hipMemsetAsync(t->cyc_block, 0, 16, s));
    }
    d.cyc_verdict = t->cyc_block;
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

// checkCycles: how many pivots the history has room for beyond what the last status poll saw (the history is grown by
// yalps_shard_poll / between the batches of yalps_shard_run, where the host knows the count: poll at least this often)
constexpr int64_t SHARD_HIST_MARGIN = 8192;

static int shard_hist_reserve(yalps_tableau *t, int64_t have) {
    if (!t->shard_check || have + SHARD_HIST_MARGIN <= t->hist_cap) return 0;
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    if (int rc = grow_history(t, have + 2 * SHARD_HIST_MARGIN, have)) return rc;
    YConst hc;
    HIP_TRY(hipMemcpy(&hc, t->d.cst, sizeof(YConst), hipMemcpyDeviceToHost));
    hc.hist_cap = t->hist_cap;
    hc.hist_leaving = t->hist[0];
    hc.hist_entering = t->hist[1];
    HIP_TRY(hipMemcpy(t->d.cst, &hc, sizeof(YConst), hipMemcpyHostToDevice));
    return 0;
}

int32_t yalps_shard_begin(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles) {
    if (!t || t->height < 1) return fail(YALPS_E_ARG, "yalps_shard_begin: no tableau uploaded");
    if (t->d.nshards < 1 || !t->d.cyc_verdict) return fail(YALPS_E_ARG, "yalps_shard_begin: yalps_tableau_set_shard first");
    HIP_TRY(hipSetDevice(t->ctx->device));
    if ((checkCycles != 0) != t->shard_check) t->generation = next_tableau_generation(); // (a captured batch has / lacks the detector's launch)
    t->shard_check = checkCycles != 0;
    if (t->shard_check && t->hist_cap < SHARD_HIST_MARGIN)
        if (int rc0 = grow_history(t, SHARD_HIST_MARGIN, 0)) return rc0;
    int rc = init_state(t, precision, maxPivots, checkCycles);
    if (rc) return rc;
    launch_one(t, 0, MODE_FUSED, 0); // bootstrap scan: emits this rank's first partials
    HIP_TRY(hipGetLastError());
    t->shard_parity = 1;
    if (t->dfn) HIP_TRY(hipMemsetAsync(t->d.dstate, 0, 2 * sizeof(DelayState), t->ctx->stream)); // nothing pending
    t->d.ext_sweep = 0; // (yalps_shard_run turns the sweep launch on: it knows the batch length)
    t->shard_pend = 0;
    if (t->wfn_inplace) // (the scan left the tableau in buffer 1 and the partials in set 1: the first in-place launch reads replica 1)
        HIP_TRY(hipMemcpyAsync(t->d.obj[1], t->d.mat[1], sizeof(double) * (size_t)t->d.pitch, hipMemcpyDeviceToDevice, t->ctx->stream));
    return 0;
}

static int shard_select_blocks(const yalps_tableau *t) { // 16-byte units of the two candidate rows over 1024-lane workgroups
    const int blocks = (t->d.pitch + 1023) / 1024;
    return blocks < 1 ? 1 : blocks > 64 ? 64 : blocks;
}

// this rank's candidates + their rows into its slot of the all-gather (delayed row updates: with the pending pivots applied)
static void launch_select(yalps_tableau *t, double *send) {
    using SelectFn = void (*)(Desc, int, double *);
    if (t->dfn) {
        const int lanes = t->nb <= 256 ? 256 : 1024; // (dshard_select_kernel<256>: four times the workgroups)
        const int blocks = std::min(256, std::max(1, (t->d.pitch + lanes - 1) / lanes));
        reinterpret_cast<SelectFn>(const_cast<void *>(yalps_dshard_select_fn(lanes)))<<<dim3(blocks), dim3(lanes), 0, t->ctx->stream>>>(t->d, t->shard_parity, send);
    }
    else
        shard_select_kernel<<<dim3(shard_select_blocks(t)), dim3(1024), 0, t->ctx->stream>>>(t->d, t->shard_parity, send);
}

int32_t yalps_shard_select(yalps_tableau *t, double *send_dev) {
    if (!t || !send_dev) return fail(YALPS_E_ARG, "yalps_shard_select: bad argument");
    launch_select(t, send_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t yalps_shard_apply(yalps_tableau *t, const double *gathered_dev) {
    if (!t || !gathered_dev) return fail(YALPS_E_ARG, "yalps_shard_apply: bad argument");
    launch_shard(t, gathered_dev);
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
    else if (int rc = shard_hist_reserve(t, now.hist_len)) return rc;
    if (status_out) *status_out = now.status;
    if (result_out) *result_out = now.result;
    if (pivots_out) *pivots_out = now.pivots;
    return 0;
}

// ---- the exchange of the row-sharded solve, natively: RCCL over xGMI --------------------------------------------
// RCCL is looked up at run time (dlopen; a copy the process already holds -- torch's -- is reused: both carry the
// soname librccl.so.1), so the library has no link-time dependency on it and single-GPU users never load it.
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, YalpsNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl *rccl_load() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break; // (already in the process?)
        if (!r.lib)
            for (const char *n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) {
            r.why = std::string("librccl.so.1 not found: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}
#define rccl() (*rccl_load())
constexpr int NCCL_FLOAT64 = 8, NCCL_UINT64 = 5, NCCL_SUM = 0; // rccl.h: ncclDataType_t / ncclRedOp_t
#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        const int e_ = (expr);                                                                                  \
        if (e_ != 0) return fail(YALPS_E_DEVICE, std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e_) : "?")); \
    } while (0)
} // namespace

struct yalps_comm {
    yalps_ctx *ctx = nullptr;
    int rank = 0, nranks = 1;
    void *nccl = nullptr;              // ncclComm_t (RCCL transport)
    yalps_allgather_fn host_fn = nullptr; // host transport (tests / hosts with their own channel): staged through pinned memory
    void *host_user = nullptr;
    double *send = nullptr, *recv = nullptr; // device slots: mine, everybody's
    double *pin = nullptr;                   // pinned staging of the host transport: send slot | gathered slots
    size_t slot_cap = 0;                     // doubles per slot the buffers hold
    hipGraphExec_t graph_exec = nullptr;     // one batch of pivots (select, all-gather, apply), captured once per tableau
    hipGraph_t graph = nullptr;
    const yalps_tableau *graph_for = nullptr;
    uint64_t graph_generation = 0; // ... and which incarnation of it (yalps_tableau::generation): the graph holds its Desc by value
    int graph_steps = 0;
    bool graph_failed = false;
    int64_t collectives = 0, graph_replays = 0;
};

int32_t yalps_comm_unique_id(void *id128) {
    if (!id128) return fail(YALPS_E_ARG, "yalps_comm_unique_id: NULL argument");
    Rccl &r = rccl();
    if (!r.lib || !r.why.empty()) return fail(YALPS_E_DEVICE, "RCCL is not usable: " + r.why);
    NCCL_TRY(r.GetUniqueId(id128));
    return 0;
}

static void comm_free(yalps_comm *c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
    delete c;
}

int32_t yalps_comm_create(yalps_ctx *ctx, const void *id128, int32_t rank, int32_t nranks, yalps_comm **out) {
    if (!ctx || !id128 || !out || nranks < 1 || nranks > MAX_SHARDS || rank < 0 || rank >= nranks)
        return fail(YALPS_E_ARG, "yalps_comm_create: bad argument");
    *out = nullptr;
    Rccl &r = rccl();
    if (!r.lib || !r.why.empty()) return fail(YALPS_E_DEVICE, "RCCL is not usable: " + r.why);
    HIP_TRY(hipSetDevice(ctx->device));
    yalps_comm *c = new yalps_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    YalpsNcclId id;
    std::memcpy(&id, id128, sizeof id);
    const int e = r.CommInitRank(&c->nccl, nranks, id, rank);
    if (e != 0) {
        comm_free(c);
        return fail(YALPS_E_DEVICE, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(e) : "?"));
    }
    *out = c;
    return 0;
}

int32_t yalps_comm_create_host(yalps_ctx *ctx, yalps_allgather_fn fn, void *user, int32_t rank, int32_t nranks, yalps_comm **out) {
    if (!ctx || !fn || !out || nranks < 1 || nranks > MAX_SHARDS || rank < 0 || rank >= nranks)
        return fail(YALPS_E_ARG, "yalps_comm_create_host: bad argument");
    yalps_comm *c = new yalps_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    c->host_fn = fn;
    c->host_user = user;
    *out = c;
    return 0;
}

void yalps_comm_destroy(yalps_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    comm_free(c);
}

static int comm_reserve(yalps_comm *c, size_t slot) {
    if (slot <= c->slot_cap) return 0;
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec); // (the captured batch holds the old addresses)
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr;
    c->graph = nullptr;
    c->graph_for = nullptr;
    if (c->send) HIP_TRY(hipFree(c->send));
    if (c->recv) HIP_TRY(hipFree(c->recv));
    if (c->pin) HIP_TRY(hipHostFree(c->pin));
    c->send = c->recv = c->pin = nullptr;
    c->slot_cap = 0;
    HIP_TRY(hipMalloc(&c->send, sizeof(double) * slot));
    HIP_TRY(hipMalloc(&c->recv, sizeof(double) * slot * (size_t)c->nranks));
    HIP_TRY(hipMemset(c->send, 0, sizeof(double) * slot));
    HIP_TRY(hipMemset(c->recv, 0, sizeof(double) * slot * (size_t)c->nranks));
    if (c->host_fn) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->pin), sizeof(double) * slot * ((size_t)c->nranks + 1), hipHostMallocDefault));
    c->slot_cap = slot;
    return 0;
}

// one pivot of the sharded solve on the context's stream: my candidates, the exchange, the elimination
static int shard_step(yalps_tableau *t, yalps_comm *c, size_t slot) {
    hipStream_t s = t->ctx->stream;
    launch_select(t, c->send);
    if (c->nccl) {
        NCCL_TRY(rccl().AllGather(c->send, c->recv, slot, NCCL_FLOAT64, c->nccl, s));
    } else { // host transport: down, the host's own all-gather, up (a test / bring-up path: one wait per pivot)
        HIP_TRY(hipMemcpyAsync(c->pin, c->send, sizeof(double) * slot, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->host_fn(c->host_user, c->pin, c->pin + slot, (int64_t)slot) != 0)
            return fail(YALPS_E_DEVICE, "yalps_shard_run: the host all-gather callback failed");
        HIP_TRY(hipMemcpyAsync(c->recv, c->pin + slot, sizeof(double) * slot * (size_t)c->nranks, hipMemcpyHostToDevice, s));
    }
    c->collectives++;
    launch_shard(t, c->recv);
    if (t->d.ext_sweep && ++t->shard_pend == t->d.delay_depth) { // `depth` pivots pending behind this step: the sweep, a launch of its own
        using SweepFn = void (*)(Desc, int);
        reinterpret_cast<SweepFn>(const_cast<void *>(t->xsweep_fn))<<<dim3(t->xsweep_grid), dim3(512), t->xsweep_lds, t->ctx->stream>>>(t->d, t->shard_parity);
        t->shard_pend = 0;
    }
    t->shard_parity ^= 1;
    return 0;
}

int32_t yalps_shard_run(yalps_tableau *t, yalps_comm *c, double precision, double maxPivots, int32_t checkCycles, int32_t check_every,
                        int32_t *status_out, double *result_out, int64_t *pivots_out, float *gpu_ms_out) {
    if (!t || !c || t->height < 1 || t->ctx != c->ctx) return fail(YALPS_E_ARG, "yalps_shard_run: bad argument");
    if (t->d.nshards != c->nranks || t->d.shard_rank != c->rank)
        return fail(YALPS_E_ARG, "yalps_shard_run: the tableau's partition (yalps_tableau_set_shard) and the communicator disagree");
    yalps_ctx *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t slot = (size_t)yalps_shard_slot_doubles(t);
    if (int rc = comm_reserve(c, slot)) return rc;
    if (check_every < 2) check_every = 2;
    if (check_every > SHARD_HIST_MARGIN / 2) check_every = (int32_t)(SHARD_HIST_MARGIN / 2);
    check_every &= ~1; // (even: the launch parity is back where it started after a batch, so one captured batch serves every replay)
    if (int rc = yalps_shard_begin(t, precision, maxPivots, checkCycles)) return rc;
    // (the host counts the pending pivots: a batch has to end with none of them pending for one captured batch to serve every replay)
    t->d.ext_sweep = (t->dfn && t->xsweep_fn && check_every % t->d.delay_depth == 0) ? 1 : 0;
    if (gpu_ms_out) HIP_TRY(hipEventRecord(ctx->ev0, s));
    // The first batch runs eagerly (RCCL sets its channels up on first use); from the second on the batch is ONE
    // hipGraph replay where the transport can be captured (RCCL's collectives can; the host transport waits per pivot).
    const bool want_graph = c->nccl && !ctx->eager && !c->graph_failed && env_int("YALPS_HIP_SHARD_GRAPH", 1);
    YState fin;
    for (int batch = 0;; batch++) {
        bool replayed = false;
        if (want_graph && batch >= 1 && !c->graph_failed) {
            if (!c->graph_exec || c->graph_for != t || c->graph_generation != t->generation || c->graph_steps != check_every) {
                if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
                if (c->graph) (void)hipGraphDestroy(c->graph);
                c->graph_exec = nullptr;
                c->graph = nullptr;
                const int parity0 = t->shard_parity;
                const int64_t coll0 = c->collectives;
                std::string why;
                hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
                bool ok = he == hipSuccess;
                if (!ok) why = std::string("hipStreamBeginCapture: ") + hipGetErrorString(he);
                int rc = 0;
                if (ok)
                    for (int i = 0; i < check_every && rc == 0; i++) rc = shard_step(t, c, slot);
                if (ok && rc != 0) why = "a step failed while being recorded: " + g_err;
                hipGraph_t g = nullptr;
                if (ok) {
                    he = hipStreamEndCapture(s, &g);
                    if (he != hipSuccess && why.empty()) why = std::string("hipStreamEndCapture: ") + hipGetErrorString(he);
                    ok = he == hipSuccess && rc == 0 && g != nullptr;
                }
                t->shard_parity = parity0; // (nothing ran: capture only recorded)
                c->collectives = coll0;
                if (ok) {
                    he = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
                    if (he != hipSuccess) why = std::string("hipGraphInstantiate: ") + hipGetErrorString(he);
                    ok = he == hipSuccess;
                }
                if (ok) {
                    c->graph = g;
                    c->graph_for = t;
                    c->graph_generation = t->generation;
                    c->graph_steps = check_every;
                } else {
                    if (g) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                    c->graph_failed = true; // (this transport / runtime cannot be captured: enqueue every pivot from here)
                    std::fprintf(stderr, "yalps_hip: the sharded pivot batch could not be captured into a hipGraph (%s); enqueueing per pivot\n", why.c_str());
                }
            }
            if (c->graph_exec) {
                HIP_TRY(hipGraphLaunch(c->graph_exec, s));
                c->collectives += check_every;
                c->graph_replays++;
                replayed = true;
            }
        }
        if (!replayed)
            for (int i = 0; i < check_every; i++)
                if (int rc = shard_step(t, c, slot)) return rc;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&t->host_state[1], t->d.st + t->shard_parity, sizeof(YState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        fin = t->host_state[1];
        if (fin.status != RUNNING) break;
        if (int rc = shard_hist_reserve(t, fin.hist_len)) return rc; // (checkCycles: room for the next batch's pivots)
    }
    if (gpu_ms_out) {
        HIP_TRY(hipEventRecord(ctx->ev1, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipEventElapsedTime(gpu_ms_out, ctx->ev0, ctx->ev1));
    }
    t->cur = fin.mbuf;
    if (fin.status == SHARD_SWEEP_MISSING) return fail(YALPS_E_DEVICE, "yalps_shard_run: a step kernel found the sweep launch missing (internal error)");
    if (status_out) *status_out = fin.status;
    if (result_out) *result_out = fin.result;
    if (pivots_out) *pivots_out = fin.pivots;
    return 0;
}

int32_t yalps_comm_info(const yalps_comm *c, char *buf, int32_t len) {
    if (!c || !buf || len < 1) return fail(YALPS_E_ARG, "yalps_comm_info: bad argument");
    std::snprintf(buf, (size_t)len, "transport=%s rank=%d nranks=%d collectives=%lld graph_replays=%lld graph=%s", c->nccl ? "rccl" : "host",
                  c->rank, c->nranks, (long long)c->collectives, (long long)c->graph_replays,
                  c->graph_exec ? "captured" : c->graph_failed ? "failed" : "none");
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
    bool lds = false; // node tableaux fit in LDS: batch_kernel<.., true>
    char *pin = nullptr; // pinned staging of batch_solve_fetch (inputs, then outputs)
    size_t pin_bytes = 0;
    void *in_block = nullptr, *out_block = nullptr; // device: cut lists | per-node results (see batch_layout)
    size_t in_bytes = 0, out_bytes = 0;
    size_t lay_in[4] = {0, 0, 0, 0}, lay_out[5] = {0, 0, 0, 0, 0}, lay_in_total = 0, lay_out_total = 0;
    int32_t last_count = 0;
};

// Lays the cut lists of `count` nodes with `total` cuts and the per-node outputs out inside the two device blocks and
// points the descriptor (and the host-side staging offsets) at them.
static void batch_layout(yalps_batch *b, size_t count, size_t total) {
    BatchDesc &d = b->d;
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    b->lay_in[0] = 0;                                        // value[total]
    b->lay_in[1] = up8(b->lay_in[0] + 8 * total);            // offsets[count + 1]
    b->lay_in[2] = up8(b->lay_in[1] + 4 * (count + 1));      // sign[total]
    b->lay_in[3] = up8(b->lay_in[2] + 4 * total);            // variable[total]
    b->lay_in_total = up8(b->lay_in[3] + 4 * total);
    char *in = static_cast<char *>(b->in_block);
    b->cut_val = reinterpret_cast<double *>(in + b->lay_in[0]);
    b->cut_off = reinterpret_cast<int32_t *>(in + b->lay_in[1]);
    b->cut_sign = reinterpret_cast<int32_t *>(in + b->lay_in[2]);
    b->cut_var = reinterpret_cast<int32_t *>(in + b->lay_in[3]);
    d.cut_val = b->cut_val;
    d.cut_off = b->cut_off;
    d.cut_sign = b->cut_sign;
    d.cut_var = b->cut_var;
    b->lay_out[0] = 0;                                                   // result[count]
    b->lay_out[1] = up8(b->lay_out[0] + 8 * count);                      // column 0: count x hmax
    b->lay_out[2] = up8(b->lay_out[1] + 8 * count * d.hmax);             // status[count]
    b->lay_out[3] = up8(b->lay_out[2] + 4 * count);                      // positionOfVariable: count x permmax
    b->lay_out[4] = up8(b->lay_out[3] + 4 * count * d.permmax);          // variableAtPosition: count x permmax
    b->lay_out_total = up8(b->lay_out[4] + 4 * count * d.permmax);
    char *out = static_cast<char *>(b->out_block);
    d.result = reinterpret_cast<double *>(out + b->lay_out[0]);
    d.ws_rhs = reinterpret_cast<double *>(out + b->lay_out[1]);
    d.status = reinterpret_cast<int32_t *>(out + b->lay_out[2]);
    d.ws_pos = reinterpret_cast<int32_t *>(out + b->lay_out[3]);
    d.ws_var = reinterpret_cast<int32_t *>(out + b->lay_out[4]);
}

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
    const int lp = small_lds_pitch(d.n);
    const size_t lds_bytes = sizeof(double) * ((size_t)d.hmax * lp + 2 * (size_t)d.hmax + (size_t)lp) +
                             sizeof(int32_t) * 2 * ((size_t)d.permmax + 1);
    b->lds = lds_bytes <= SMALL_LDS_MAX && !env_int("YALPS_HIP_NO_LDS", 0);
    b->shmem = b->lds ? lds_bytes : sizeof(double) * ((size_t)d.hmax + d.pitch);
    if (b->shmem > 150 * 1024) return fail(YALPS_E_ARG, "yalps_batch_create: node tableau too large for the batched path");
    const size_t nm = (size_t)max_nodes;
    HIP_TRY(hipMalloc(&b->root_mat, sizeof(double) * (size_t)d.h0 * d.pitch));
    HIP_TRY(hipMemset(b->root_mat, 0, sizeof(double) * (size_t)d.h0 * d.pitch));
    HIP_TRY(hipMalloc(&b->root_rhs, sizeof(double) * (size_t)d.h0));
    HIP_TRY(hipMalloc(&b->root_pos, sizeof(int32_t) * (size_t)(width + d.h0)));
    HIP_TRY(hipMalloc(&b->root_var, sizeof(int32_t) * (size_t)(width + d.h0)));
    HIP_TRY(hipMalloc(&d.ws_mat, sizeof(double) * nm * d.hmax * d.pitch));
    // inputs (cut lists) and outputs (status, result, column 0, permutations of every node) each in ONE block, so
    // that a batch costs one copy up and one copy down (batch_solve_fetch lays them out tightly per batch)
    b->in_bytes = sizeof(int32_t) * (nm + 1) + 16 * nm * (size_t)max_cuts + 64;
    b->out_bytes = nm * (sizeof(int32_t) + sizeof(double) + sizeof(double) * d.hmax + 2 * sizeof(int32_t) * d.permmax) + 64;
    HIP_TRY(hipMalloc(&b->in_block, b->in_bytes));
    HIP_TRY(hipMalloc(&b->out_block, b->out_bytes));
    batch_layout(b, max_nodes, max_nodes * max_cuts);
    HIP_TRY(hipMalloc(&d.height, sizeof(int32_t) * nm));
    HIP_TRY(hipMalloc(&d.pivots, sizeof(long long) * nm));
    d.root_mat = b->root_mat;
    d.root_rhs = b->root_rhs;
    d.root_pos = b->root_pos;
    d.root_var = b->root_var;
    if (b->shmem > 48 * 1024)
        if (int rc = allow_big_lds(ctx->device, b->lds ? reinterpret_cast<const void *>(batch_kernel<256, true>)
                                                       : reinterpret_cast<const void *>(batch_kernel<1024, false>)))
            return rc;
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
    void *bufs[] = {b->root_mat, b->root_rhs, b->root_pos, b->root_var, b->d.ws_mat, b->in_block, b->out_block, b->d.height, b->d.pivots};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (b->pin) (void)hipHostFree(b->pin);
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
    // everything the kernel will index with is checked here, before anything is enqueued (like apply_cuts_impl)
    if (cut_offsets[0] != 0) return fail(YALPS_E_ARG, "yalps_batch_solve: cut_offsets[0] must be 0");
    for (int32_t i = 0; i < count; i++)
        if (cut_offsets[i + 1] < cut_offsets[i] || cut_offsets[i + 1] - cut_offsets[i] > b->max_cuts)
            return fail(YALPS_E_ARG, "yalps_batch_solve: cut_offsets must not decrease, and a node has at most max_cuts cuts");
    const int32_t total = cut_offsets[count];
    if ((int64_t)total > (int64_t)b->max_nodes * b->max_cuts) return fail(YALPS_E_ARG, "yalps_batch_solve: more cuts than the batch has room for");
    for (int32_t i = 0; i < total; i++)
        if (cut_var[i] < 0 || cut_var[i] >= b->d.w + b->d.h0)
            return fail(YALPS_E_ARG, "yalps_batch_solve: cut on an unknown variable");
    yalps_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    batch_layout(b, (size_t)b->max_nodes, (size_t)b->max_nodes * b->max_cuts); // (the capacity layout)
    HIP_TRY(hipMemcpyAsync(b->cut_off, cut_offsets, sizeof(int32_t) * (size_t)(count + 1), hipMemcpyHostToDevice, s));
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(b->cut_sign, cut_sign, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->cut_var, cut_var, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->cut_val, cut_value, sizeof(double) * (size_t)total, hipMemcpyHostToDevice, s));
    }
    b->d.precision = precision;
    b->d.max_pivots = maxPivots;
    HIP_TRY(hipEventRecord(c->ev0, s));
    if (b->lds)
        batch_kernel<256, true><<<dim3(count), dim3(256), b->shmem, s>>>(b->d);
    else
        // nodes in the HBM workspace are >= 150 KB each: 1024 lanes (4+ row groups) stream them at 5.6 TB/s, 256 lanes at 3.9
        batch_kernel<1024, false><<<dim3(count), dim3(1024), b->shmem, s>>>(b->d);
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

// One batch for the native branch-and-cut driver: cuts up, kernel, and per node status / result / column 0 / both
// permutations back -- everything through one pinned staging buffer, ONE wait.  col0_all / pos_all / var_all are
// count x hmax doubles and count x permmax int32 (node i at i * hmax / i * permmax).
static int32_t batch_solve_fetch(yalps_batch *b, int32_t count, const int32_t *off, const int32_t *sign, const int32_t *var,
                                 const double *val, double precision, double maxPivots, int32_t *status_out, double *result_out,
                                 double *col0_all, int32_t *pos_all, int32_t *var_all) {
    yalps_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const BatchDesc &d = b->d;
    const size_t total = (size_t)off[count], n = (size_t)count;
    batch_layout(b, n, total); // tight: one copy up, one copy down
    const size_t bytes = b->lay_in_total + b->lay_out_total;
    if (bytes > b->pin_bytes) {
        if (b->pin) HIP_TRY(hipHostFree(b->pin));
        b->pin = nullptr;
        b->pin_bytes = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->pin), 2 * bytes, hipHostMallocDefault));
        b->pin_bytes = 2 * bytes;
    }
    char *pin_in = b->pin, *pin_out = b->pin + b->lay_in_total;
    std::memcpy(pin_in + b->lay_in[0], val, 8 * total);
    std::memcpy(pin_in + b->lay_in[1], off, 4 * (n + 1));
    std::memcpy(pin_in + b->lay_in[2], sign, 4 * total);
    std::memcpy(pin_in + b->lay_in[3], var, 4 * total);
    HIP_TRY(hipMemcpyAsync(b->in_block, pin_in, b->lay_in_total, hipMemcpyHostToDevice, s));
    b->d.precision = precision;
    b->d.max_pivots = maxPivots;
    if (b->lds)
        batch_kernel<256, true><<<dim3(count), dim3(256), b->shmem, s>>>(b->d);
    else
        batch_kernel<1024, false><<<dim3(count), dim3(1024), b->shmem, s>>>(b->d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pin_out, b->out_block, b->lay_out_total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(result_out, pin_out + b->lay_out[0], 8 * n);
    std::memcpy(col0_all, pin_out + b->lay_out[1], 8 * n * d.hmax);
    std::memcpy(status_out, pin_out + b->lay_out[2], 4 * n);
    std::memcpy(pos_all, pin_out + b->lay_out[3], 4 * n * d.permmax);
    std::memcpy(var_all, pin_out + b->lay_out[4], 4 * n * d.permmax);
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

// the process-wide tableau behind the host-array entry points (g_default_mu held by the caller)
int default_ctx() {
    if (g_default_ctx) return 0;
    return yalps_ctx_create(env_int("YALPS_HIP_DEVICE", 0), &g_default_ctx);
}

int default_tableau(int32_t width, int32_t height, yalps_tableau **out) {
    int rc = default_ctx();
    if (rc) return rc;
    yalps_tableau *t = g_default_tab;
    if (!t || t->d.w != width || t->d.hcap < height || t->d.hcap > 4 * height) {
        if (t) yalps_tableau_destroy(t);
        g_default_tab = nullptr;
        rc = yalps_tableau_create(g_default_ctx, width, height, &t);
        if (rc) return rc;
        g_default_tab = t;
    }
    *out = t;
    return 0;
}
} // namespace

// (g_default_mu held by the caller: the public entry point below, or yalps_milp_f64 for its host-side nodes)
static int32_t simplex_f64_locked(double *matrix, int32_t width, int32_t height, int32_t *pos, int32_t *var,
                                  double precision, double maxPivots, int32_t checkCycles, int32_t copyback,
                                  double *result_out, int64_t *pivots_out) {
    yalps_tableau *t = nullptr;
    int rc = default_ctx();
    if (rc) return rc;
    if (fits_small(g_default_ctx, width, height)) {
        // small tableau: staged in pinned host memory in the reference's own layout; the kernel reads
        // and writes it there over PCIe (no copies to HBM, one launch, one synchronisation)
        yalps_ctx *c = g_default_ctx;
        HIP_TRY(hipSetDevice(c->device));
        const size_t nm = (size_t)width * height, np = (size_t)width + height;
        const size_t bytes = sizeof(double) * nm + sizeof(int32_t) * 2 * np;
        if (bytes > c->small_blob_cap) {
            if (c->small_blob) HIP_TRY(hipHostFree(c->small_blob));
            c->small_blob = nullptr;
            c->small_blob_cap = 0;
            const size_t cap = bytes < (64u << 10) ? (64u << 10) : 2 * bytes;
            HIP_TRY(hipHostMalloc(&c->small_blob, cap, hipHostMallocDefault));
            c->small_blob_cap = cap;
        }
        double *bm = static_cast<double *>(c->small_blob);
        int32_t *bp = reinterpret_cast<int32_t *>(bm + nm), *bv = bp + np;
        std::memcpy(bm, matrix, sizeof(double) * nm);
        std::memcpy(bp, pos, sizeof(int32_t) * np);
        std::memcpy(bv, var, sizeof(int32_t) * np);
        SmallDesc sd{};
        sd.mat = bm + 1;
        sd.rhs = bm;
        sd.pitch = width;
        sd.rhs_stride = width;
        sd.pos = bp;
        sd.var = bv;
        sd.w = width;
        sd.n = width - 1;
        sd.h = height;
        sd.precision = precision;
        sd.max_pivots = maxPivots;
        const int32_t status = run_small(c, sd, checkCycles, result_out, pivots_out, nullptr);
        if (status < 0) return status;
        if (copyback == YALPS_COPYBACK_SOLUTION)
            for (int32_t r = 0; r < height; r++) matrix[(size_t)r * width] = bm[(size_t)r * width];
        else
            std::memcpy(matrix, bm, sizeof(double) * nm);
        std::memcpy(pos, bp, sizeof(int32_t) * np);
        std::memcpy(var, bv, sizeof(int32_t) * np);
        return status;
    }
    rc = default_tableau(width, height, &t);
    if (rc) return rc;
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

int32_t yalps_simplex_f64_ex(double *matrix, int32_t width, int32_t height, int32_t *pos, int32_t *var,
                             double precision, double maxPivots, int32_t checkCycles, int32_t copyback,
                             double *result_out, int64_t *pivots_out) {
    if (!matrix || !pos || !var || width < 1 || height < 1)
        return fail(YALPS_E_ARG, "yalps_simplex_f64: bad argument");
    std::lock_guard<std::mutex> lock(g_default_mu);
    return simplex_f64_locked(matrix, width, height, pos, var, precision, maxPivots, checkCycles, copyback, result_out, pivots_out);
}

int32_t yalps_simplex_sparse_f64(int32_t width, int32_t height, int64_t nnz, const int32_t *row, const int32_t *col,
                                 const double *val, double precision, double maxPivots, int32_t checkCycles,
                                 double *col0_out, int32_t *pos_out, int32_t *var_out, double *result_out,
                                 int64_t *pivots_out) {
    if (width < 1 || height < 1 || !col0_out || !pos_out || !var_out)
        return fail(YALPS_E_ARG, "yalps_simplex_sparse_f64: bad argument");
    std::lock_guard<std::mutex> lock(g_default_mu);
    yalps_tableau *t = nullptr;
    int rc = default_tableau(width, height, &t);
    if (rc) return rc;
    rc = yalps_tableau_assemble(t, height, nnz, row, col, val);
    if (rc) return rc;
    const int32_t status = yalps_tableau_solve(t, precision, maxPivots, checkCycles, result_out, pivots_out, nullptr);
    if (status < 0) return status;
    rc = yalps_tableau_download_rhs(t, col0_out);
    if (rc) return rc;
    rc = yalps_tableau_download(t, nullptr, pos_out, var_out);
    if (rc) return rc;
    return status;
}

int32_t yalps_simplex_f64(double *matrix, int32_t width, int32_t height, int32_t *pos, int32_t *var,
                          double precision, double maxPivots, int32_t checkCycles, double *result_out) {
    return yalps_simplex_f64_ex(matrix, width, height, pos, var, precision, maxPivots, checkCycles,
                                YALPS_COPYBACK_FULL, result_out, nullptr);
}

} // extern "C"

#include "milp_host.inc"
