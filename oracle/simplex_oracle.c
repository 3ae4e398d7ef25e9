/*
 * oracle/simplex_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar, single-threaded CPU restatement (plain C99) of the dense-tableau
 * two-phase simplex of Ivordir/YALPS.  It exists only as the checker for the
 * HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  Nothing under yalps_amd/ imports, links or executes it.
 *
 * Parity status: PINNED.  tests/golden/ holds pivot-by-pivot traces, final
 * permutations, final RHS columns and whole-matrix digests produced by the
 * reference's own src/simplex.ts (type-erased, run under node 12 in the build
 * container by oracle/tools/gen_golden.py); tests/test_oracle_golden.py
 * requires this file to reproduce every one of them bit for bit.
 *
 * Reference lines restated (all paths are /root/reference/...):
 *   data layout      src/tableau.ts:9-21     (row-major Float64Array, row 0 =
 *                                             objective, column 0 = RHS)
 *   pivot            src/simplex.ts:5-39
 *   hasCycle         src/simplex.ts:44-63
 *   phase2           src/simplex.ts:66-103
 *   phase1           src/simplex.ts:106-142  (exported as `simplex`, :144)
 *   roundToPrecision src/util.ts:1-4
 *
 * Build with -O2 -ffp-contract=off -fno-fast-math: the reference runs on V8,
 * which never contracts a - b*c into an fma and always divides for real.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define YO_OPTIMAL 0
#define YO_INFEASIBLE 1
#define YO_UNBOUNDED 2
#define YO_CYCLED 3

typedef struct {
    double *m;
    int32_t w, h;
    int32_t *pos; /* positionOfVariable */
    int32_t *var; /* variableAtPosition */
    int32_t *nz;  /* scratch: nonZeroColumns, w entries */
    /* optional pivot trace (test instrumentation only) */
    int32_t *trace_rows, *trace_cols;
    int64_t trace_cap, n_pivots;
} yo_tab;

typedef struct {
    int32_t *leaving, *entering;
    int64_t len, cap;
} yo_hist;

#ifdef _OPENMP
#include <omp.h>
/* threads of the row-parallel build (0 = leave OpenMP's default); returns the count in force */
int32_t yalps_oracle_set_threads(int32_t n) {
    if (n > 0) omp_set_num_threads(n);
    return (int32_t)omp_get_max_threads();
}
#else
int32_t yalps_oracle_set_threads(int32_t n) {
    (void)n;
    return 1;
}
#endif

/* JS Math.round: nearest integer, halves toward +infinity (src/util.ts:2-3). */
static double js_round(double x) {
    if (!(x == x) || isinf(x)) return x;
    double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

/* src/util.ts:1-4 */
double yalps_oracle_round_to_precision(double num, double precision) {
    double rounding = js_round(1.0 / precision);
    return js_round((num + 2.220446049250313e-16) * rounding) / rounding;
}

/* src/simplex.ts:5-39 */
static void yo_pivot(yo_tab *t, int32_t row, int32_t col) {
    const int32_t w = t->w, h = t->h;
    double *m = t->m;
    double *prow = m + (size_t)row * w;
    const double quotient = prow[col];
    /* :7-12 basis bookkeeping */
    const int32_t leaving = t->var[w + row];
    const int32_t entering = t->var[col];
    t->var[w + row] = entering;
    t->var[col] = leaving;
    t->pos[leaving] = col;
    t->pos[entering] = w + row;

    /* :14-24 normalise the pivot row, flushing |v| <= 1e-16 to +0.0 */
    int32_t nnz = 0;
    for (int32_t c = 0; c < w; c++) {
        const double value = prow[c];
        if (fabs(value) > 1e-16) {
            prow[c] = value / quotient;
            t->nz[nnz++] = c;
        } else {
            prow[c] = 0.0;
        }
    }
    prow[col] = 1.0 / quotient; /* :25 */

    /* :27-38 eliminate the pivot column from every other row.  (liboracle_omp.so, -fopenmp: the rows are independent,
     * so the row loop may be split over threads without changing one bit -- the all-core CPU baseline of BASELINE.md
     * section 4.2; liboracle.so is built without OpenMP and ignores the pragma.) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (h >= 64)
#endif
    for (int32_t r = 0; r < h; r++) {
        if (r == row) continue;
        double *mr = m + (size_t)r * w;
        const double coef = mr[col];
        if (fabs(coef) > 1e-16) {
            for (int32_t i = 0; i < nnz; i++) {
                const int32_t c = t->nz[i];
                const double prod = coef * prow[c]; /* rounded product ... */
                mr[c] = mr[c] - prod;               /* ... then rounded difference */
            }
            mr[col] = -coef / quotient;
        }
    }
    if (t->trace_rows && t->n_pivots < t->trace_cap) {
        t->trace_rows[t->n_pivots] = row;
        t->trace_cols[t->n_pivots] = col;
    }
    t->n_pivots++;
}

/* src/simplex.ts:44-63 */
static int yo_has_cycle(yo_hist *hs, const yo_tab *t, int32_t row, int32_t col) {
    if (hs->len == hs->cap) {
        hs->cap = hs->cap ? hs->cap * 2 : 1024;
        hs->leaving = (int32_t *)realloc(hs->leaving, sizeof(int32_t) * (size_t)hs->cap);
        hs->entering = (int32_t *)realloc(hs->entering, sizeof(int32_t) * (size_t)hs->cap);
    }
    hs->leaving[hs->len] = t->var[t->w + row];
    hs->entering[hs->len] = t->var[col];
    hs->len++;
    for (int64_t length = 6; length <= hs->len / 2; length++) {
        int cycle = 1;
        for (int64_t i = 0; i < length; i++) {
            const int64_t item = hs->len - 1 - i;
            if (hs->leaving[item] != hs->leaving[item - length] ||
                hs->entering[item] != hs->entering[item - length]) {
                cycle = 0;
                break;
            }
        }
        if (cycle) return 1;
    }
    return 0;
}

/* src/simplex.ts:66-103 */
static int32_t yo_phase2(yo_tab *t, double precision, double maxPivots, int checkCycles,
                         double *result) {
    const int32_t w = t->w, h = t->h;
    const double *m = t->m;
    yo_hist hs = {0, 0, 0, 0};
    int32_t status = YO_CYCLED;
    *result = NAN;
    for (double iter = 0; iter < maxPivots; iter++) {
        /* :71-79 Dantzig pricing, strict >, first wins */
        int32_t col = 0;
        double value = precision;
        for (int32_t c = 1; c < w; c++) {
            const double reducedCost = m[c];
            if (reducedCost > value) {
                value = reducedCost;
                col = c;
            }
        }
        if (col == 0) { /* :80 */
            status = YO_OPTIMAL;
            *result = yalps_oracle_round_to_precision(m[0], precision);
            break;
        }
        /* :83-95 min-ratio test with the early break */
        int32_t row = 0;
        double minRatio = INFINITY;
        for (int32_t r = 1; r < h; r++) {
            const double v = m[(size_t)r * w + col];
            if (v <= precision) continue;
            const double rhs = m[(size_t)r * w];
            const double ratio = rhs / v;
            if (ratio < minRatio) {
                row = r;
                minRatio = ratio;
                if (ratio <= precision) break;
            }
        }
        if (row == 0) { /* :96 */
            status = YO_UNBOUNDED;
            *result = (double)col;
            break;
        }
        if (checkCycles && yo_has_cycle(&hs, t, row, col)) break; /* :98 */
        yo_pivot(t, row, col);                                      /* :100 */
    }
    free(hs.leaving);
    free(hs.entering);
    return status;
}

/* src/simplex.ts:106-142 */
static int32_t yo_phase1(yo_tab *t, double precision, double maxPivots, int checkCycles,
                         double *result) {
    const int32_t w = t->w, h = t->h;
    const double *m = t->m;
    yo_hist hs = {0, 0, 0, 0};
    int32_t status = YO_CYCLED;
    *result = NAN;
    for (double iter = 0; iter < maxPivots; iter++) {
        /* :111-119 most negative RHS, strict <, first wins */
        int32_t row = 0;
        double rhs = -precision;
        for (int32_t r = 1; r < h; r++) {
            const double value = m[(size_t)r * w];
            if (value < rhs) {
                rhs = value;
                row = r;
            }
        }
        if (row == 0) { /* :120 */
            free(hs.leaving);
            free(hs.entering);
            return yo_phase2(t, precision, maxPivots, checkCycles, result);
        }
        /* :123-134 entering column by max ratio, strict >, first wins */
        int32_t col = 0;
        double maxRatio = -INFINITY;
        for (int32_t c = 1; c < w; c++) {
            const double coefficient = m[(size_t)row * w + c];
            if (coefficient < -precision) {
                const double ratio = -m[c] / coefficient;
                if (ratio > maxRatio) {
                    maxRatio = ratio;
                    col = c;
                }
            }
        }
        if (col == 0) { /* :135 */
            status = YO_INFEASIBLE;
            break;
        }
        if (checkCycles && yo_has_cycle(&hs, t, row, col)) break; /* :137 */
        yo_pivot(t, row, col);                                      /* :139 */
    }
    free(hs.leaving);
    free(hs.entering);
    return status;
}

/*
 * Same argument list as the product entry point yalps_simplex_f64
 * (include/yalps_hip.h) plus an optional pivot trace.
 * Returns the status code; *result_out follows src/simplex.ts's return
 * protocol: optimal -> rounded M[0,0]; unbounded -> entering column index;
 * infeasible / cycled -> NaN.
 */
int32_t yalps_oracle_simplex_f64(double *matrix, int32_t width, int32_t height,
                                 int32_t *positionOfVariable, int32_t *variableAtPosition,
                                 double precision, double maxPivots, int32_t checkCycles,
                                 double *result_out, int32_t *trace_rows, int32_t *trace_cols,
                                 int64_t trace_cap, int64_t *n_pivots_out) {
    yo_tab t;
    t.m = matrix;
    t.w = width;
    t.h = height;
    t.pos = positionOfVariable;
    t.var = variableAtPosition;
    t.nz = (int32_t *)malloc(sizeof(int32_t) * (size_t)(width > 0 ? width : 1));
    t.trace_rows = trace_rows;
    t.trace_cols = trace_cols;
    t.trace_cap = trace_cap;
    t.n_pivots = 0;
    double result = NAN;
    int32_t status = yo_phase1(&t, precision, maxPivots, checkCycles != 0, &result);
    free(t.nz);
    if (result_out) *result_out = result;
    if (n_pivots_out) *n_pivots_out = t.n_pivots;
    return status;
}

/* One bare pivot (src/simplex.ts:5-39), for kernel-level parity tests. */
void yalps_oracle_pivot_f64(double *matrix, int32_t width, int32_t height,
                            int32_t *positionOfVariable, int32_t *variableAtPosition, int32_t row,
                            int32_t col) {
    yo_tab t;
    memset(&t, 0, sizeof t);
    t.m = matrix;
    t.w = width;
    t.h = height;
    t.pos = positionOfVariable;
    t.var = variableAtPosition;
    t.nz = (int32_t *)malloc(sizeof(int32_t) * (size_t)(width > 0 ? width : 1));
    yo_pivot(&t, row, col);
    free(t.nz);
}

/*
 * dense-LP(M,N,seed) of SURVEY.md section 8(d): the reference test-suite's
 * PRNG (tests/helpers/util.ts:20-41, prospector hash; the seed is an IEEE
 * double incremented by 0x9e3779b9 with no wrap) filling a (M+1)x(N+1)
 * tableau: c_j into row 0, then per row b_r = N*0.25*(1+rand()) and A_rj.
 * An independent restatement of the generator the product ships in
 * yalps_amd/csrc (yalps_dense_lp_f64); the golden traces pin both.
 */
static uint32_t yo_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0xd35a2d97u;
    x ^= x >> 15;
    return x;
}

static double yo_rand(double *seed) {
    *seed += 2654435769.0; /* 0x9e3779b9, in double arithmetic */
    /* JS ToInt32: the integer value modulo 2^32.  An integer-valued seed below 2^63 converts exactly to uint64_t and
     * the truncation to 32 bits IS the modulo (fmod, 50 x slower, only for anything else). */
    uint32_t x;
    if (*seed >= 0.0 && *seed < 9223372036854775808.0 && *seed == (double)(uint64_t)*seed)
        x = (uint32_t)(uint64_t)*seed;
    else
        x = (uint32_t)(uint64_t)fmod(*seed, 4294967296.0);
    return (double)yo_hash32(x) / 4294967296.0;
}

void yalps_oracle_dense_lp_f64(int32_t M, int32_t N, double seed, double *matrix) {
    const int32_t w = N + 1, h = M + 1;
    matrix[0] = 0.0;
    for (int32_t j = 1; j < w; j++) matrix[j] = yo_rand(&seed);
    for (int32_t r = 1; r < h; r++) {
        double *mr = matrix + (size_t)r * w;
        mr[0] = (double)N * 0.25 * (1.0 + yo_rand(&seed));
        for (int32_t j = 1; j < w; j++) mr[j] = yo_rand(&seed);
    }
}
