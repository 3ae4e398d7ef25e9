"""ctypes binding of oracle/liboracle.so (the CPU checker).  Test-side only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
STATUS = ("optimal", "infeasible", "unbounded", "cycled")

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.yalps_oracle_simplex_f64.restype = C.c_int32
        lib.yalps_oracle_simplex_f64.argtypes = [_f64p, C.c_int32, C.c_int32, _i32p, _i32p, C.c_double, C.c_double,
                                                 C.c_int32, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.POINTER(C.c_int64)]
        lib.yalps_oracle_pivot_f64.restype = None
        lib.yalps_oracle_pivot_f64.argtypes = [_f64p, C.c_int32, C.c_int32, _i32p, _i32p, C.c_int32, C.c_int32]
        lib.yalps_oracle_dense_lp_f64.restype = None
        lib.yalps_oracle_dense_lp_f64.argtypes = [C.c_int32, C.c_int32, C.c_double, _f64p]
        lib.yalps_oracle_round_to_precision.restype = C.c_double
        lib.yalps_oracle_round_to_precision.argtypes = [C.c_double, C.c_double]
        lib.yalps_oracle_set_threads.restype = C.c_int32
        lib.yalps_oracle_set_threads.argtypes = [C.c_int32]

    def set_threads(self, n):
        """Row-parallel build only (load(omp=True)); returns the thread count in force (1 for the scalar build)."""
        return self.lib.yalps_oracle_set_threads(int(n))

    def simplex(self, matrix, width, height, pos, var, precision=1e-8, max_pivots=8192.0, check_cycles=False,
                trace_cap=0):
        """In-place like the reference's simplex(); returns (status, result, n_pivots, trace[n,2])."""
        assert matrix.dtype == np.float64 and matrix.size == width * height
        res, npiv = C.c_double(), C.c_int64()
        tr = np.zeros(max(trace_cap, 1), np.int32)
        tc = np.zeros(max(trace_cap, 1), np.int32)
        st = self.lib.yalps_oracle_simplex_f64(matrix, width, height, pos, var, precision, float(max_pivots),
                                               int(bool(check_cycles)), C.byref(res),
                                               tr.ctypes.data if trace_cap else None,
                                               tc.ctypes.data if trace_cap else None, trace_cap, C.byref(npiv))
        n = min(npiv.value, trace_cap)
        return STATUS[st], res.value, npiv.value, np.stack([tr[:n], tc[:n]], axis=1)

    def pivot(self, matrix, width, height, pos, var, row, col):
        self.lib.yalps_oracle_pivot_f64(matrix, width, height, pos, var, row, col)

    def dense_lp(self, M, N, seed=42.0):
        m = np.zeros((M + 1) * (N + 1), np.float64)
        self.lib.yalps_oracle_dense_lp_f64(M, N, float(seed), m)
        return m

    def round_to_precision(self, x, precision):
        return self.lib.yalps_oracle_round_to_precision(x, precision)


def build(name="liboracle.so"):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, name], check=True)
    return os.path.join(ORACLE_DIR, name)


def load(omp=False):
    """omp=True: liboracle_omp.so, the same source with the elimination's row loop split over threads (-fopenmp)."""
    name = "liboracle_omp.so" if omp else "liboracle.so"
    path = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "simplex_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(name)
    return Oracle(C.CDLL(path))
