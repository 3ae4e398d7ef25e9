/*
 * yalps_hip.h -- C ABI of the MI355X (gfx950) dense-tableau simplex core.
 *
 * This is the drop-in boundary for ONE hot path of Ivordir/YALPS: the ES-module
 * export `simplex` (reference src/simplex.ts:144; called at src/YALPS.ts:79 and
 * src/branchAndCut.ts:127).  The reference has no FFI of its own; the entry
 * points below are exactly what an N-API / ctypes binding of that export needs
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only.
 *
 * Data layout at the boundary = reference src/tableau.ts:9-21: `matrix` is a
 * row-major Float64Array of width*height doubles, row 0 = objective row,
 * column 0 = RHS column; `positionOfVariable` / `variableAtPosition` are Int32
 * arrays of width+height entries.  Everything is mutated in place exactly as
 * src/simplex.ts does.
 *
 * Return protocol (src/simplex.ts:66-69,80,96,98,102,120,135,137,141):
 *   return value >= 0 : SolutionStatus code below; *result_out = rounded M[0,0]
 *                       (optimal) | entering column index (unbounded) | NaN.
 *   return value <  0 : native failure (HIP error, OOM, bad argument) -- the
 *                       reference never throws on this path, so this is the
 *                       out-of-band channel; text via yalps_last_error().
 * There is NO CPU fallback: without a usable gfx950 device every entry point
 * fails with YALPS_E_DEVICE.
 *
 * Threads: the host-array entry points (yalps_simplex_f64[_ex], yalps_simplex_sparse_f64, yalps_milp_f64) share one
 * process-wide context and serialise on one mutex.  A yalps_ctx and the objects created from it belong to one thread
 * at a time (their calls enqueue on the context's stream and share its scratch); different contexts are independent
 * (solves that use the whole chip in one persistent launch take turns per device inside the library).
 * yalps_last_error() is per thread.
 */
#ifndef YALPS_HIP_H
#define YALPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SolutionStatus codes produced by the hot path (src/types.ts SolutionStatus;
 * "timedout" is produced only by branch-and-cut, never by simplex()). */
#define YALPS_OPTIMAL 0
#define YALPS_INFEASIBLE 1
#define YALPS_UNBOUNDED 2
#define YALPS_CYCLED 3
#define YALPS_TIMEDOUT 4 /* only from yalps_milp_f64 (branch and cut: timeout / maxIterations, src/branchAndCut.ts:166-173) */

#define YALPS_E_ARG (-1)    /* bad argument */
#define YALPS_E_DEVICE (-2) /* no usable HIP device / HIP runtime error */
#define YALPS_E_NOMEM (-3)  /* device or host allocation failed */

/* What yalps_simplex_f64_ex copies back to the caller's arrays. */
#define YALPS_COPYBACK_FULL 0     /* whole matrix + both permutations (reference semantics) */
#define YALPS_COPYBACK_SOLUTION 1 /* column 0 (strided into matrix) + both permutations: all that
                                     solution() reads for a pure LP (src/YALPS.ts:18-19,32) */

const char *yalps_last_error(void);
int32_t yalps_device_count(void);

/* ---- the drop-in: replaces `simplex(tableau, options)` (src/simplex.ts:106-144) -------------
 * Host arrays in, host arrays out, synchronous, non-re-entrant (like the single JS thread).
 * Options read on this path: precision, maxPivots (may be +Infinity), checkCycles
 * (src/simplex.ts:68,108).  The arrays are not retained after return. */
int32_t yalps_simplex_f64(double *matrix, int32_t width, int32_t height, int32_t *positionOfVariable,
                          int32_t *variableAtPosition, double precision, double maxPivots,
                          int32_t checkCycles, double *result_out);

int32_t yalps_simplex_f64_ex(double *matrix, int32_t width, int32_t height,
                             int32_t *positionOfVariable, int32_t *variableAtPosition,
                             double precision, double maxPivots, int32_t checkCycles,
                             int32_t copyback, double *result_out, int64_t *pivots_out);

/* tableauModel -> simplex -> solution for a pure LP without a dense host tableau: the initial
 * tableau arrives as its written cells (see yalps_tableau_assemble), and only what solution()
 * reads comes back (src/YALPS.ts:18-19,32): column 0 (height doubles) and both permutations
 * (width+height int32 each). */
int32_t yalps_simplex_sparse_f64(int32_t width, int32_t height, int64_t nnz, const int32_t *row,
                                 const int32_t *col, const double *val, double precision,
                                 double maxPivots, int32_t checkCycles, double *col0_out,
                                 int32_t *pos_out, int32_t *var_out, double *result_out,
                                 int64_t *pivots_out);

/* ---- device-resident tableaux (HBM) -------------------------------------------------------
 * For callers that keep the tableau on the GPU between calls (benchmarks, branch-and-cut node
 * evaluation: src/branchAndCut.ts:126-127 re-solves root+cuts per node). */
typedef struct yalps_ctx yalps_ctx;
typedef struct yalps_tableau yalps_tableau;

int32_t yalps_ctx_create(int32_t device, yalps_ctx **out);
/* Same, but every kernel / copy is enqueued on the caller's HIP stream (e.g. the stream a
 * torch.distributed collective is ordered against). */
int32_t yalps_ctx_create_on_stream(int32_t device, void *hip_stream, yalps_ctx **out);
void yalps_ctx_destroy(yalps_ctx *ctx);

/* A tableau of `width` columns and room for up to `height_capacity` rows (any width: rows wider than 16385 columns
 * are solved by an untuned any-shape kernel pair). */
int32_t yalps_tableau_create(yalps_ctx *ctx, int32_t width, int32_t height_capacity,
                             yalps_tableau **out);
void yalps_tableau_destroy(yalps_tableau *t);

/* Host -> HBM: row-major width*height doubles + the two permutations (width+height int32). */
int32_t yalps_tableau_upload(yalps_tableau *t, const double *matrix, int32_t height,
                             const int32_t *positionOfVariable, const int32_t *variableAtPosition);
/* On-device assembly of an INITIAL tableau (what src/tableau.ts:87-134 builds):
 * rows [0,height) are zeroed in HBM, the `nnz` cells (row[i], col[i]) = val[i] are scattered
 * (column 0 = RHS column, row 0 = objective row) and both permutations are set to the identity
 * (src/tableau.ts:95-98).  Cells must be sorted by (row, col), strictly increasing -- the host
 * resolves the reference's "a later coefficient overwrites an earlier one" (:101-115) before the
 * call.  Moves 16*nnz bytes over PCIe instead of 8*width*height. */
int32_t yalps_tableau_assemble(yalps_tableau *t, int32_t height, int64_t nnz, const int32_t *row,
                               const int32_t *col, const double *val);
/* HBM -> host; any of the three pointers may be NULL to skip it. */
int32_t yalps_tableau_download(yalps_tableau *t, double *matrix, int32_t *positionOfVariable,
                               int32_t *variableAtPosition);
/* HBM -> host, only column 0 (height doubles, contiguous). */
int32_t yalps_tableau_download_rhs(yalps_tableau *t, double *col0);
/* HBM -> host, what solution() / mostFractionalVar read: column 0 (height doubles) and both permutations, one wait. */
int32_t yalps_tableau_download_solution(yalps_tableau *t, double *col0, int32_t *positionOfVariable,
                                        int32_t *variableAtPosition);
/* HBM -> HBM copy of matrix + permutations + height (same width; dst capacity >= src height). */
int32_t yalps_tableau_copy(yalps_tableau *dst, const yalps_tableau *src);
/* applyCuts (src/branchAndCut.ts:22-61) on the device: dst = root's tableau (HBM -> HBM) + one row per cut
 * (sign, variable, value), with the new slack variables appended to both permutations.  `root` holds the root's
 * OPTIMAL tableau (after its own solve) and is left untouched; dst needs capacity for root height + ncuts rows.
 * A branch-and-cut node then costs no PCIe traffic beyond its cuts, column 0 and the permutations. */
int32_t yalps_tableau_apply_cuts(yalps_tableau *dst, const yalps_tableau *root, int32_t ncuts,
                                 const int32_t *cut_sign, const int32_t *cut_variable, const double *cut_value);
/* One branch-and-cut node in one call (src/branchAndCut.ts:126-127 `applyCuts` + `simplex` on the node, then what
 * `mostFractionalVar` / `solution()` read of it): dst = root + cuts as yalps_tableau_apply_cuts builds it, solved, column 0
 * (root height + ncuts doubles) and both permutations (width + root height + ncuts entries) copied back when the node is
 * optimal.  Where the node's tableau takes the register-resident kernel the whole sequence is three kernel launches
 * and one wait (node_prepare_kernel: copies + cuts + state, the solve, node_finish_kernel: results into pinned host
 * memory) instead of eleven stream operations; otherwise it is the same calls one by one.  Returns the SolutionStatus
 * code or a negative error. */
int32_t yalps_tableau_node_solve(yalps_tableau *dst, const yalps_tableau *root, int32_t ncuts, const int32_t *cut_sign,
                                 const int32_t *cut_variable, const double *cut_value, double precision, double maxPivots,
                                 int32_t checkCycles, double *result_out, double *col0_out, int32_t *positionOfVariable_out,
                                 int32_t *variableAtPosition_out);
int32_t yalps_tableau_height(const yalps_tableau *t);
/* Which kernels this tableau uses and which path the last solve took (text, for benchmarks). */
int32_t yalps_tableau_info(const yalps_tableau *t, char *buf, int32_t len);
/* Diagnostic build only (library compiled with -DYALPS_STAMPS; tools/resident_stages.py): per-workgroup sums of
 * in-kernel stage stamps of the persistent launches since the last reset, 24 uint64 words per workgroup (stage sums in
 * shader cycles [0..19], pivots [20], s_memtime span [21], s_memrealtime span [22]).  Returns the number of words
 * copied; the shipped library executes no stamp and returns YALPS_E_ARG. */
int32_t yalps_tableau_debug_stamps(yalps_tableau *t, uint64_t *out, int32_t cap_words, int32_t reset);
/* Diagnostic: device rows are padded to a multiple of 16 doubles; counts the padding doubles of the current buffer that
 * are not finite / not zero (tests: a kernel that multiplied a padding lane by a marker would show here). */
int32_t yalps_tableau_padding_check(yalps_tableau *t, int64_t *nonfinite_out, int64_t *nonzero_out);

/* Run the two-phase simplex on the resident tableau (in place).  gpu_ms_out (optional) =
 * HIP-event time of the whole pivot loop on the context's stream. */
int32_t yalps_tableau_solve(yalps_tableau *t, double precision, double maxPivots,
                            int32_t checkCycles, double *result_out, int64_t *pivots_out,
                            float *gpu_ms_out);

/* One bare Gauss-Jordan pivot (src/simplex.ts:5-39) on the resident tableau. */
int32_t yalps_tableau_pivot(yalps_tableau *t, int32_t row, int32_t col);

/* Measurement hook (bench.py `roofline.onchip_floor`): `epochs` rounds of the register-resident kernels' per-pivot exchange
 * and nothing else -- `workgroups` x `lanes`, one per CU, rows of 2 * units * lanes doubles; variant bit 0: every workgroup
 * publishes a row (write-through, drained) before its 16-byte record, bit 1: everybody fetches the winner's row afterwards,
 * bit 2: the two-level exchange (the first workgroup of every XCD gathers its XCD's records and raises one record per XCD).
 * *us_per_epoch_out = HIP-event time / epochs.  (lanes x units: 512x2 = the 16 KB rows of BASELINE config 2, 512x3, 256x1, 256x2.) */
int32_t yalps_ctx_exchange_floor(yalps_ctx *ctx, int32_t workgroups, int32_t lanes, int32_t units, int32_t epochs, int32_t variant,
                                 float *us_per_epoch_out);

/* Measurement hook: `launches` back-to-back launches of the row-elimination sweep kernel alone
 * (pivot (row,col) re-applied each time; the tableau content is consumed), timed with HIP
 * events on the context's stream.  *avg_us_out = average kernel duration. */
int32_t yalps_tableau_bench_sweep(yalps_tableau *t, int32_t row, int32_t col, int32_t launches,
                                  float *avg_us_out);

/* ---- row-sharded solve of ONE tableau across GPUs (BASELINE config 5) ------------
 * One process per GPU.  Each rank uploads a local tableau = objective row + its contiguous block
 * of rows, declares the partition with yalps_tableau_set_shard (bounds[r]..bounds[r+1] = global
 * rows of rank r, bounds[0] = 1, bounds[nranks] = global height; pos/var are the GLOBAL
 * permutations, width + global height entries), then runs per pivot:
 *     yalps_shard_select(t, send)            -- this rank's candidates + candidate rows
 *     all-gather of yalps_shard_slot_doubles() doubles per rank   (caller: RCCL / torch.distributed)
 *     yalps_shard_apply(t, gathered)         -- every rank picks the same winner and eliminates
 * All calls only enqueue work on the context's stream; yalps_shard_poll synchronises and returns
 * the replicated status (-1 = still running).  checkCycles (src/simplex.ts:44-63,98,137): one more single-workgroup
 * launch per pivot runs the detector on the pivot everybody is about to decide (permutations and history are replicated
 * on every rank: no communication); the history grows inside yalps_shard_poll -- poll at least every 8192 pivots. */
int32_t yalps_tableau_set_shard(yalps_tableau *t, int32_t rank, int32_t nranks, const int32_t *bounds,
                                int32_t global_height, const int32_t *positionOfVariable,
                                const int32_t *variableAtPosition);
int64_t yalps_shard_slot_doubles(const yalps_tableau *t);
int32_t yalps_shard_begin(yalps_tableau *t, double precision, double maxPivots, int32_t checkCycles);
int32_t yalps_shard_select(yalps_tableau *t, double *send_dev);
int32_t yalps_shard_apply(yalps_tableau *t, const double *gathered_dev);
int32_t yalps_shard_poll(yalps_tableau *t, int32_t *status_out, double *result_out, int64_t *pivots_out);

/* The same loop run natively, exchange included: no host-language call per pivot.  A yalps_comm is this rank's end of
 * the exchange: RCCL over xGMI (looked up at run time; yalps_comm_unique_id on rank 0 yields the 128 bytes that every
 * rank passes to yalps_comm_create, carried between the processes by the host's own channel), or -- for hosts with a
 * channel of their own, and for tests -- a host callback that all-gathers `doubles_per_rank` doubles per rank between
 * host buffers (send_host: mine; recv_host: nranks slots in rank order; returns 0).
 * yalps_shard_run = yalps_shard_begin + { select, all-gather, apply } until the replicated status is final, the status
 * read back every `check_every` pivots; with RCCL the batch of check_every pivots is captured once into a hipGraph and
 * replayed (a check_every that is a multiple of the shard's delay depth -- 16 or 8; 64 is -- lets shards with few rows per
 * workgroup sweep their rows in a launch of their own, dsweep_kernel.cuh; any other value works, with the sweep inside the step
 * kernel).  Returns 0 or a negative error; *status_out / *result_out follow src/simplex.ts's return protocol and are
 * identical on every rank. */
typedef struct yalps_comm yalps_comm;
typedef int32_t (*yalps_allgather_fn)(void *user, const double *send_host, double *recv_host, int64_t doubles_per_rank);
int32_t yalps_comm_unique_id(void *id128);
int32_t yalps_comm_create(yalps_ctx *ctx, const void *id128, int32_t rank, int32_t nranks, yalps_comm **out);
int32_t yalps_comm_create_host(yalps_ctx *ctx, yalps_allgather_fn fn, void *user, int32_t rank, int32_t nranks,
                               yalps_comm **out);
void yalps_comm_destroy(yalps_comm *c);
int32_t yalps_comm_info(const yalps_comm *c, char *buf, int32_t len);
int32_t yalps_shard_run(yalps_tableau *t, yalps_comm *c, double precision, double maxPivots, int32_t checkCycles, int32_t check_every,
                        int32_t *status_out, double *result_out, int64_t *pivots_out, float *gpu_ms_out);

/* ---- batched branch-and-cut node evaluation (BASELINE config 4) ---------------
 * src/branchAndCut.ts:126-127 evaluates every node as simplex(applyCuts(root, cuts)); nodes are
 * independent given the root.  A batch keeps the root's optimal tableau resident, builds each
 * node's tableau on the device (applyCuts, :22-61: one row per cut (sign, variable, value)) and
 * solves `count` nodes concurrently, one workgroup per node.  Cuts of node i are entries
 * [cut_offsets[i], cut_offsets[i+1]) of the three cut arrays.  checkCycles is not available here. */
typedef struct yalps_batch yalps_batch;
int32_t yalps_batch_create(yalps_ctx *ctx, int32_t width, int32_t root_height, int32_t max_cuts,
                           int32_t max_nodes, yalps_batch **out);
void yalps_batch_destroy(yalps_batch *b);
/* Host -> HBM: the root tableau AFTER its own simplex() (row-major width*root_height) + permutations. */
int32_t yalps_batch_set_root(yalps_batch *b, const double *matrix, const int32_t *positionOfVariable,
                             const int32_t *variableAtPosition);
int32_t yalps_batch_solve(yalps_batch *b, int32_t count, const int32_t *cut_offsets, const int32_t *cut_sign,
                          const int32_t *cut_variable, const double *cut_value, double precision,
                          double maxPivots, int32_t *status_out, double *result_out, int64_t *pivots_out,
                          float *gpu_ms_out);
/* Node `node` of the last batch, `height` = root_height + its number of cuts; NULL pointers are skipped. */
int32_t yalps_batch_download(yalps_batch *b, int32_t node, int32_t height, double *matrix, double *col0,
                             int32_t *positionOfVariable, int32_t *variableAtPosition);

/* ---- the whole branch and cut in one native call (src/YALPS.ts:73-92 for a model with integers) ----------
 * Input: the INITIAL tableau as tableauModel built it (row-major, identity permutations allowed but not required),
 * the 1-based columns of the integer variables (TableauModel.integers, src/tableau.ts:57-71), `sign` (-1 minimise,
 * +1 maximise, :51) and the options read on this path (src/types.ts:203-265; timeout in ms, may be +Infinity).
 * Does simplex() on the root, then branchAndCut (src/branchAndCut.ts:89-176) with the reference's queue order
 * (binary heap with heapq / heap.js sift rules, keyed by the parent's evaluation) and every node LP on the GPU:
 * node_batch > 1 evaluates that many frontier nodes per launch (speculatively; results are committed in pop
 * order), otherwise one node at a time with the root resident in HBM.
 * Output: *status_out (YALPS_* incl. YALPS_TIMEDOUT), *result_out (best evaluation or NaN; root result when the
 * root is not optimal), and what solution() reads of the best tableau: *height_out, column 0 (col0_out, room for
 * height + 2*n_integers doubles) and both permutations (room for width + height + 2*n_integers entries).
 * stats_out (optional, 3 x int64): nodes consumed, node LPs evaluated, batches.  Returns 0 or a negative error. */
int32_t yalps_milp_f64(const double *matrix, int32_t width, int32_t height, const int32_t *positionOfVariable,
                       const int32_t *variableAtPosition, const int32_t *integers, int32_t n_integers, double sign,
                       double precision, double maxPivots, int32_t checkCycles, double tolerance, double timeout_ms,
                       double maxIterations, int32_t node_batch, int32_t *status_out, double *result_out,
                       double *col0_out, int32_t *pos_out, int32_t *var_out, int32_t *height_out, int64_t *stats_out);

/* ---- synthetic input of the headline benchmark ---------------------------------------------
 * dense-LP(M,N,seed): fills a (M+1) x (N+1) row-major tableau with the
 * reference test-suite's PRNG stream (tests/helpers/util.ts:20-41).  Host-side, deterministic. */
void yalps_dense_lp_f64(int32_t M, int32_t N, double seed, double *matrix);
/* Rows [row_begin, row_end) of the same tableau (row 0 = objective row), row-major into `rows`: what one rank of a
 * row-sharded solve holds -- the stream is walked through the rows before, nothing else is generated or stored. */
void yalps_dense_lp_rows_f64(int32_t M, int32_t N, double seed, int32_t row_begin, int32_t row_end, double *rows);

/* JS-exact roundToPrecision (src/util.ts:1-4), for host marshalling code. */
double yalps_round_to_precision(double num, double precision);

#ifdef __cplusplus
}
#endif
#endif /* YALPS_HIP_H */
