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
    if (!g_simplex || !g_last_error) {
        g_load_error = path + " does not export yalps_simplex_f64 / yalps_last_error";
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

napi_value Init(napi_env env, napi_value exports) {
    napi_value fn;
    napi_create_function(env, "simplex", NAPI_AUTO_LENGTH, Simplex, nullptr, &fn);
    napi_set_named_property(env, exports, "simplex", fn);
    return exports;
}

} // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
