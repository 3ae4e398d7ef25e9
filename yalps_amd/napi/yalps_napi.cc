// yalps_napi.cc -- thin N-API shim over the C ABI of include/yalps_hip.h.
//
// Exposes `simplex(tableau, options) -> [status, number]` with exactly the signature of the
// reference's ES-module export (src/simplex.ts:144, `(tableau: Tableau, options:
// Required<Options>) => [SolutionStatus, number]`), so that a maintainer can replace the body of
// src/simplex.ts by a re-export of this addon.  No computation happens here:
// the three typed arrays of the Tableau object (src/tableau.ts:9-15) are handed to
// yalps_simplex_f64 in place.  Typed-array VIEWS are honoured (byte offset / length): branch and
// cut passes subarray() views shorter than their buffers (src/branchAndCut.ts:55-59).
//
// For branch and cut (src/branchAndCut.ts:89-176) three more exports keep the root's optimal tableau in HBM and
// build every node there (yalps_tableau_apply_cuts), so that a node costs no tableau transfer:
//   rootSolve(tableau, options, maxCuts) -> [status, number, handle]   simplex() on the root; the tableau's
//        column 0 and both permutations are written back (what mostFractionalVar :64-85 / solution() read)
//   nodeSolve(handle, cuts, options, out) -> [status, number, height]  applyCuts (:22-61) + simplex (:127) on the
//        device; cuts = [[sign, variable, value], ...]; out = {col0: Float64Array, positionOfVariable, variableAtPosition}
//   rootFree(handle)
// and, one step further, the whole of it in one native call (yalps_milp_f64: native heap, nodes in GPU batches):
//   solveInteger(tableau, integers, sign, options, nodeBatch) -> [status, number, height]
//        tableau = the INITIAL tableau; on return its column 0 (first `height` rows of a matrix that must have room
//        for height + 2*integers.length rows) and both permutations are those of the best tableau
//
// libyalps_hip.so is loaded with dlopen from YALPS_HIP_LIB or next to this addon; a missing
// library or GPU surfaces as a thrown JS Error (the reference itself never throws on this path,
// so this is the out-of-band channel for native failures).
#include <dlfcn.h>
#include <node_api.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>

namespace {

using simplex_fn = int32_t (*)(double *, int32_t, int32_t, int32_t *, int32_t *, double, double, int32_t, double *);
using last_error_fn = const char *(*)();

simplex_fn g_simplex = nullptr;
last_error_fn g_last_error = nullptr;
// handle-based entry points (include/yalps_hip.h), resolved together with the two above
int32_t (*g_ctx_create)(int32_t, void **) = nullptr;
void (*g_ctx_destroy)(void *) = nullptr;
int32_t (*g_tab_create)(void *, int32_t, int32_t, void **) = nullptr;
void (*g_tab_destroy)(void *) = nullptr;
int32_t (*g_tab_upload)(void *, const double *, int32_t, const int32_t *, const int32_t *) = nullptr;
int32_t (*g_tab_download)(void *, double *, int32_t *, int32_t *) = nullptr;
int32_t (*g_tab_download_rhs)(void *, double *) = nullptr;
int32_t (*g_tab_height)(const void *) = nullptr;
int32_t (*g_tab_solve)(void *, double, double, int32_t, double *, int64_t *, float *) = nullptr;
int32_t (*g_tab_apply_cuts)(void *, const void *, int32_t, const int32_t *, const int32_t *, const double *) = nullptr;
int32_t (*g_tab_node_solve)(void *, const void *, int32_t, const int32_t *, const int32_t *, const double *, double, double, int32_t,
                            double *, double *, int32_t *, int32_t *) = nullptr;
int32_t (*g_milp)(const double *, int32_t, int32_t, const int32_t *, const int32_t *, const int32_t *, int32_t, double, double, double,
                  int32_t, double, double, double, int32_t, int32_t *, double *, double *, int32_t *, int32_t *, int32_t *,
                  int64_t *) = nullptr;
std::string g_load_error;

bool load_library() {
    if (g_simplex) return true;
    std::string path;
    if (const char *env = std::getenv("YALPS_HIP_LIB")) {
        path = env;
    } else {
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&load_library), &info) && info.dli_fname) {
            path = info.dli_fname;
            const size_t slash = path.rfind('/');
            path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/../libyalps_hip.so";
        }
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        g_load_error = std::string("cannot load ") + path + ": " + dlerror();
        return false;
    }
    g_simplex = reinterpret_cast<simplex_fn>(dlsym(h, "yalps_simplex_f64"));
    g_last_error = reinterpret_cast<last_error_fn>(dlsym(h, "yalps_last_error"));
    auto sym = [&](const char *name) { return dlsym(h, name); };
    g_ctx_create = reinterpret_cast<decltype(g_ctx_create)>(sym("yalps_ctx_create"));
    g_ctx_destroy = reinterpret_cast<decltype(g_ctx_destroy)>(sym("yalps_ctx_destroy"));
    g_tab_create = reinterpret_cast<decltype(g_tab_create)>(sym("yalps_tableau_create"));
    g_tab_destroy = reinterpret_cast<decltype(g_tab_destroy)>(sym("yalps_tableau_destroy"));
    g_tab_upload = reinterpret_cast<decltype(g_tab_upload)>(sym("yalps_tableau_upload"));
    g_tab_download = reinterpret_cast<decltype(g_tab_download)>(sym("yalps_tableau_download"));
    g_tab_download_rhs = reinterpret_cast<decltype(g_tab_download_rhs)>(sym("yalps_tableau_download_rhs"));
    g_tab_height = reinterpret_cast<decltype(g_tab_height)>(sym("yalps_tableau_height"));
    g_tab_solve = reinterpret_cast<decltype(g_tab_solve)>(sym("yalps_tableau_solve"));
    g_tab_apply_cuts = reinterpret_cast<decltype(g_tab_apply_cuts)>(sym("yalps_tableau_apply_cuts"));
    g_tab_node_solve = reinterpret_cast<decltype(g_tab_node_solve)>(sym("yalps_tableau_node_solve"));
    g_milp = reinterpret_cast<decltype(g_milp)>(sym("yalps_milp_f64"));
    if (!g_simplex || !g_last_error || !g_ctx_create || !g_ctx_destroy || !g_tab_create || !g_tab_destroy || !g_tab_upload ||
        !g_tab_download || !g_tab_download_rhs || !g_tab_height || !g_tab_solve || !g_tab_apply_cuts || !g_tab_node_solve || !g_milp) {
        g_load_error = path + " does not export the yalps_* entry points of include/yalps_hip.h";
        g_simplex = nullptr;
        return false;
    }
    return true;
}

napi_value fail(napi_env env, const std::string &msg) {
    napi_throw_error(env, nullptr, msg.c_str());
    return nullptr;
}

bool get_named(napi_env env, napi_value obj, const char *name, napi_value *out) {
    bool has = false;
    return napi_has_named_property(env, obj, name, &has) == napi_ok && has &&
           napi_get_named_property(env, obj, name, out) == napi_ok;
}

bool get_typed(napi_env env, napi_value obj, const char *name, napi_typedarray_type want, void **data,
               size_t *length) {
    napi_value v;
    if (!get_named(env, obj, name, &v)) return false;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_typedarray_type type;
    napi_value ab;
    size_t byte_offset;
    // `data` already points at the first element of the VIEW (buffer base + byte_offset)
    if (napi_get_typedarray_info(env, v, &type, length, data, &ab, &byte_offset) != napi_ok) return false;
    return type == want;
}

// simplex(tableau, options) -> [status, number]
napi_value Simplex(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 2)
        return fail(env, "simplex(tableau, options): two arguments expected");
    if (!load_library()) return fail(env, g_load_error);

    void *matrix = nullptr, *pos = nullptr, *var = nullptr;
    size_t nmat = 0, npos = 0, nvar = 0;
    if (!get_typed(env, argv[0], "matrix", napi_float64_array, &matrix, &nmat) ||
        !get_typed(env, argv[0], "positionOfVariable", napi_int32_array, &pos, &npos) ||
        !get_typed(env, argv[0], "variableAtPosition", napi_int32_array, &var, &nvar))
        return fail(env, "tableau must have matrix: Float64Array and positionOfVariable / variableAtPosition: Int32Array");
    napi_value v;
    int32_t width = 0, height = 0;
    if (!get_named(env, argv[0], "width", &v) || napi_get_value_int32(env, v, &width) != napi_ok ||
        !get_named(env, argv[0], "height", &v) || napi_get_value_int32(env, v, &height) != napi_ok)
        return fail(env, "tableau.width / tableau.height must be numbers");
    if (width < 1 || height < 1 || nmat < (size_t)width * (size_t)height || npos < (size_t)width + (size_t)height ||
        nvar < (size_t)width + (size_t)height)
        return fail(env, "tableau arrays are shorter than width/height imply");

    double precision = 1e-8, max_pivots = 8192;
    bool check_cycles = false;
    if (get_named(env, argv[1], "precision", &v)) napi_get_value_double(env, v, &precision);
    if (get_named(env, argv[1], "maxPivots", &v)) napi_get_value_double(env, v, &max_pivots); // may be Infinity
    if (get_named(env, argv[1], "checkCycles", &v)) napi_get_value_bool(env, v, &check_cycles);

    double result = NAN;
    const int32_t status = g_simplex(static_cast<double *>(matrix), width, height, static_cast<int32_t *>(pos),
                                     static_cast<int32_t *>(var), precision, max_pivots, check_cycles ? 1 : 0,
                                     &result);
    if (status < 0) return fail(env, std::string("yalps_hip: ") + g_last_error());

    static const char *const kStatus[] = {"optimal", "infeasible", "unbounded", "cycled"};
    napi_value out, s, r;
    napi_create_array_with_length(env, 2, &out);
    napi_create_string_utf8(env, kStatus[status], NAPI_AUTO_LENGTH, &s);
    napi_create_double(env, result, &r);
    napi_set_element(env, out, 0, s);
    napi_set_element(env, out, 1, r);
    return out;
}

struct Options {
    double precision = 1e-8, max_pivots = 8192;
    bool check_cycles = false;
};

Options read_options(napi_env env, napi_value obj) {
    Options o;
    napi_value v;
    if (get_named(env, obj, "precision", &v)) napi_get_value_double(env, v, &o.precision);
    if (get_named(env, obj, "maxPivots", &v)) napi_get_value_double(env, v, &o.max_pivots); // may be Infinity
    if (get_named(env, obj, "checkCycles", &v)) napi_get_value_bool(env, v, &o.check_cycles);
    return o;
}

napi_value status_tuple(napi_env env, int32_t status, double result, napi_value third) {
    static const char *const kStatus[] = {"optimal", "infeasible", "unbounded", "cycled"};
    napi_value out, s, r;
    napi_create_array_with_length(env, third ? 3 : 2, &out);
    napi_create_string_utf8(env, kStatus[status], NAPI_AUTO_LENGTH, &s);
    napi_create_double(env, result, &r);
    napi_set_element(env, out, 0, s);
    napi_set_element(env, out, 1, r);
    if (third) napi_set_element(env, out, 2, third);
    return out;
}

// The root of a branch and cut kept in HBM: context, root tableau, one node tableau (root height + maxCuts rows).
struct Root {
    void *ctx = nullptr, *root = nullptr, *node = nullptr;
    int32_t width = 0, height = 0, max_cuts = 0;
    ~Root() {
        if (node) g_tab_destroy(node);
        if (root) g_tab_destroy(root);
        if (ctx) g_ctx_destroy(ctx);
    }
};

void finalize_root(napi_env, void *data, void *) { delete static_cast<Root *>(data); }

// rootSolve(tableau, options, maxCuts) -> [status, number, handle]
napi_value RootSolve(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 3)
        return fail(env, "rootSolve(tableau, options, maxCuts): three arguments expected");
    if (!load_library()) return fail(env, g_load_error);
    void *matrix = nullptr, *pos = nullptr, *var = nullptr;
    size_t nmat = 0, npos = 0, nvar = 0;
    if (!get_typed(env, argv[0], "matrix", napi_float64_array, &matrix, &nmat) ||
        !get_typed(env, argv[0], "positionOfVariable", napi_int32_array, &pos, &npos) ||
        !get_typed(env, argv[0], "variableAtPosition", napi_int32_array, &var, &nvar))
        return fail(env, "tableau must have matrix: Float64Array and positionOfVariable / variableAtPosition: Int32Array");
    napi_value v;
    int32_t width = 0, height = 0, max_cuts = 0;
    if (!get_named(env, argv[0], "width", &v) || napi_get_value_int32(env, v, &width) != napi_ok ||
        !get_named(env, argv[0], "height", &v) || napi_get_value_int32(env, v, &height) != napi_ok ||
        napi_get_value_int32(env, argv[2], &max_cuts) != napi_ok || max_cuts < 0)
        return fail(env, "tableau.width / tableau.height / maxCuts must be numbers");
    if (width < 1 || height < 1 || nmat < (size_t)width * (size_t)height || npos < (size_t)width + (size_t)height ||
        nvar < (size_t)width + (size_t)height)
        return fail(env, "tableau arrays are shorter than width/height imply");
    const Options o = read_options(env, argv[1]);
    Root *r = new Root();
    r->width = width;
    r->height = height;
    r->max_cuts = max_cuts;
    double result = NAN;
    int32_t status = g_ctx_create(0, &r->ctx);
    if (status >= 0) status = g_tab_create(r->ctx, width, height, &r->root);
    if (status >= 0) status = g_tab_create(r->ctx, width, height + max_cuts, &r->node);
    if (status >= 0)
        status = g_tab_upload(r->root, static_cast<double *>(matrix), height, static_cast<int32_t *>(pos), static_cast<int32_t *>(var));
    if (status >= 0) status = g_tab_solve(r->root, o.precision, o.max_pivots, o.check_cycles ? 1 : 0, &result, nullptr, nullptr);
    const int32_t solved = status;
    if (status >= 0) { // column 0 (strided into the matrix) + both permutations back into the caller's tableau
        std::string col0((size_t)height * sizeof(double), '\0');
        double *c0 = reinterpret_cast<double *>(&col0[0]);
        status = g_tab_download_rhs(r->root, c0);
        if (status >= 0) status = g_tab_download(r->root, nullptr, static_cast<int32_t *>(pos), static_cast<int32_t *>(var));
        if (status >= 0)
            for (int32_t i = 0; i < height; i++) static_cast<double *>(matrix)[(size_t)i * width] = c0[i];
    }
    if (status < 0) {
        delete r;
        return fail(env, std::string("yalps_hip: ") + g_last_error());
    }
    napi_value handle;
    napi_create_external(env, r, finalize_root, nullptr, &handle);
    return status_tuple(env, solved, result, handle);
}

// nodeSolve(handle, cuts, options, out) -> [status, number, height]
napi_value NodeSolve(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 4)
        return fail(env, "nodeSolve(handle, cuts, options, out): four arguments expected");
    void *data = nullptr;
    if (napi_get_value_external(env, argv[0], &data) != napi_ok || !data) return fail(env, "nodeSolve: bad handle");
    Root *r = static_cast<Root *>(data);
    uint32_t ncuts = 0;
    bool is_array = false;
    if (napi_is_array(env, argv[1], &is_array) != napi_ok || !is_array || napi_get_array_length(env, argv[1], &ncuts) != napi_ok)
        return fail(env, "nodeSolve: cuts must be an array of [sign, variable, value]");
    if ((int64_t)ncuts > r->max_cuts) return fail(env, "nodeSolve: more cuts than rootSolve reserved rows for");
    std::string sbuf((size_t)ncuts * sizeof(int32_t) + 4, '\0'), vbuf((size_t)ncuts * sizeof(int32_t) + 4, '\0'),
        dbuf((size_t)ncuts * sizeof(double) + 8, '\0');
    int32_t *sign = reinterpret_cast<int32_t *>(&sbuf[0]), *variable = reinterpret_cast<int32_t *>(&vbuf[0]);
    double *value = reinterpret_cast<double *>(&dbuf[0]);
    for (uint32_t i = 0; i < ncuts; i++) {
        napi_value cut, e;
        double s = 0, vr = 0;
        if (napi_get_element(env, argv[1], i, &cut) != napi_ok || napi_get_element(env, cut, 0, &e) != napi_ok ||
            napi_get_value_double(env, e, &s) != napi_ok || napi_get_element(env, cut, 1, &e) != napi_ok ||
            napi_get_value_double(env, e, &vr) != napi_ok || napi_get_element(env, cut, 2, &e) != napi_ok ||
            napi_get_value_double(env, e, &value[i]) != napi_ok)
            return fail(env, "nodeSolve: a cut is not [sign, variable, value]");
        sign[i] = (int32_t)s;
        variable[i] = (int32_t)vr;
    }
    const Options o = read_options(env, argv[2]);
    void *col0 = nullptr, *pos = nullptr, *var = nullptr;
    size_t ncol0 = 0, npos = 0, nvar = 0;
    const size_t h = (size_t)r->height + ncuts;
    if (!get_typed(env, argv[3], "col0", napi_float64_array, &col0, &ncol0) ||
        !get_typed(env, argv[3], "positionOfVariable", napi_int32_array, &pos, &npos) ||
        !get_typed(env, argv[3], "variableAtPosition", napi_int32_array, &var, &nvar) || ncol0 < h ||
        npos < (size_t)r->width + h || nvar < (size_t)r->width + h)
        return fail(env, "nodeSolve: out must hold col0 (height) and both permutations (width + height)");
    double result = NAN;
    // applyCuts + simplex + column 0 / permutations of an optimal node in one native call (three launches, one wait)
    int32_t status = g_tab_node_solve(r->node, r->root, (int32_t)ncuts, sign, variable, value, o.precision, o.max_pivots,
                                      o.check_cycles ? 1 : 0, &result, static_cast<double *>(col0), static_cast<int32_t *>(pos),
                                      static_cast<int32_t *>(var));
    const int32_t solved = status;
    if (status > 0) { // (not optimal: the caller may still look at the tableau it was left with)
        status = g_tab_download_rhs(r->node, static_cast<double *>(col0));
        if (status >= 0) status = g_tab_download(r->node, nullptr, static_cast<int32_t *>(pos), static_cast<int32_t *>(var));
    }
    if (status < 0) return fail(env, std::string("yalps_hip: ") + g_last_error());
    napi_value hv;
    napi_create_int32(env, (int32_t)h, &hv);
    return status_tuple(env, solved, result, hv);
}

// rootFree(handle): releases the HBM copies now (otherwise the garbage collector does)
napi_value RootFree(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    void *data = nullptr;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 1 ||
        napi_get_value_external(env, argv[0], &data) != napi_ok || !data)
        return fail(env, "rootFree(handle)");
    Root *r = static_cast<Root *>(data);
    if (r->node) g_tab_destroy(r->node);
    if (r->root) g_tab_destroy(r->root);
    if (r->ctx) g_ctx_destroy(r->ctx);
    r->node = r->root = r->ctx = nullptr;
    napi_value undef;
    napi_get_undefined(env, &undef);
    return undef;
}

// solveInteger(tableau, integers, sign, options, nodeBatch) -> [status, number, height]
napi_value SolveInteger(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 5)
        return fail(env, "solveInteger(tableau, integers, sign, options, nodeBatch): five arguments expected");
    if (!load_library()) return fail(env, g_load_error);
    void *matrix = nullptr, *pos = nullptr, *var = nullptr, *ints = nullptr;
    size_t nmat = 0, npos = 0, nvar = 0, nints = 0;
    napi_value v;
    int32_t width = 0, height = 0, node_batch = 0;
    double sign = 1.0;
    if (!get_typed(env, argv[0], "matrix", napi_float64_array, &matrix, &nmat) ||
        !get_typed(env, argv[0], "positionOfVariable", napi_int32_array, &pos, &npos) ||
        !get_typed(env, argv[0], "variableAtPosition", napi_int32_array, &var, &nvar) ||
        !get_named(env, argv[0], "width", &v) || napi_get_value_int32(env, v, &width) != napi_ok ||
        !get_named(env, argv[0], "height", &v) || napi_get_value_int32(env, v, &height) != napi_ok)
        return fail(env, "solveInteger: tableau must be a Tableau (src/tableau.ts:9-15)");
    bool is_typed = false;
    napi_typedarray_type tt;
    napi_value ab;
    size_t off = 0;
    if (napi_is_typedarray(env, argv[1], &is_typed) != napi_ok || !is_typed ||
        napi_get_typedarray_info(env, argv[1], &tt, &nints, &ints, &ab, &off) != napi_ok || tt != napi_int32_array)
        return fail(env, "solveInteger: integers must be an Int32Array of 1-based tableau columns");
    if (napi_get_value_double(env, argv[2], &sign) != napi_ok || napi_get_value_int32(env, argv[4], &node_batch) != napi_ok)
        return fail(env, "solveInteger: sign / nodeBatch must be numbers");
    const size_t hmax = (size_t)height + 2 * nints;
    if (width < 1 || height < 1 || nmat < (size_t)width * (size_t)height || npos < (size_t)width + hmax || nvar < (size_t)width + hmax)
        return fail(env, "solveInteger: the permutations need room for width + height + 2*integers.length entries");
    const Options o = read_options(env, argv[3]);
    double tolerance = 0.0, timeout = INFINITY, max_iterations = 32768;
    if (get_named(env, argv[3], "tolerance", &v)) napi_get_value_double(env, v, &tolerance);
    if (get_named(env, argv[3], "timeout", &v)) napi_get_value_double(env, v, &timeout);
    if (get_named(env, argv[3], "maxIterations", &v)) napi_get_value_double(env, v, &max_iterations);
    std::string col0(hmax * sizeof(double), '\0');
    double *c0 = reinterpret_cast<double *>(&col0[0]);
    int32_t status = 0, best_height = 0;
    double result = NAN;
    const int32_t rc = g_milp(static_cast<double *>(matrix), width, height, static_cast<int32_t *>(pos), static_cast<int32_t *>(var),
                              static_cast<int32_t *>(ints), (int32_t)nints, sign, o.precision, o.max_pivots, o.check_cycles ? 1 : 0,
                              tolerance, timeout, max_iterations, node_batch, &status, &result, c0, static_cast<int32_t *>(pos),
                              static_cast<int32_t *>(var), &best_height, nullptr);
    if (rc < 0) return fail(env, std::string("yalps_hip: ") + g_last_error());
    const size_t rows = nmat / (size_t)width; // column 0 of as many rows as the caller's matrix holds
    for (size_t r = 0; r < (size_t)best_height && r < rows; r++) static_cast<double *>(matrix)[r * width] = c0[r];
    static const char *const kStatus[] = {"optimal", "infeasible", "unbounded", "cycled", "timedout"};
    napi_value out, s2, r2, h2;
    napi_create_array_with_length(env, 3, &out);
    napi_create_string_utf8(env, kStatus[status], NAPI_AUTO_LENGTH, &s2);
    napi_create_double(env, result, &r2);
    napi_create_int32(env, best_height, &h2);
    napi_set_element(env, out, 0, s2);
    napi_set_element(env, out, 1, r2);
    napi_set_element(env, out, 2, h2);
    return out;
}

napi_value Init(napi_env env, napi_value exports) {
    napi_value fn;
    napi_create_function(env, "simplex", NAPI_AUTO_LENGTH, Simplex, nullptr, &fn);
    napi_set_named_property(env, exports, "simplex", fn);
    napi_create_function(env, "rootSolve", NAPI_AUTO_LENGTH, RootSolve, nullptr, &fn);
    napi_set_named_property(env, exports, "rootSolve", fn);
    napi_create_function(env, "nodeSolve", NAPI_AUTO_LENGTH, NodeSolve, nullptr, &fn);
    napi_set_named_property(env, exports, "nodeSolve", fn);
    napi_create_function(env, "rootFree", NAPI_AUTO_LENGTH, RootFree, nullptr, &fn);
    napi_set_named_property(env, exports, "rootFree", fn);
    napi_create_function(env, "solveInteger", NAPI_AUTO_LENGTH, SolveInteger, nullptr, &fn);
    napi_set_named_property(env, exports, "solveInteger", fn);
    return exports;
}

} // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
