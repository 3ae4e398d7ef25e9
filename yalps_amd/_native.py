"""ctypes binding of libyalps_hip.so (include/yalps_hip.h).

There is no CPU path: if the library is missing, or no gfx950 device is usable,
every call raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YALPS_HIP_LIB") or os.path.join(HERE, "libyalps_hip.so")  # (same switch as the N-API addon)

STATUS = ("optimal", "infeasible", "unbounded", "cycled", "timedout")
COPYBACK_FULL, COPYBACK_SOLUTION = 0, 1

# every symbol include/yalps_hip.h declares
SYMBOLS = (
    "yalps_last_error", "yalps_device_count", "yalps_simplex_f64", "yalps_simplex_f64_ex", "yalps_ctx_create",
    "yalps_ctx_destroy", "yalps_tableau_create", "yalps_tableau_destroy", "yalps_tableau_upload",
    "yalps_tableau_download", "yalps_tableau_download_rhs", "yalps_tableau_copy", "yalps_tableau_height",
    "yalps_tableau_solve", "yalps_tableau_pivot", "yalps_tableau_bench_sweep", "yalps_ctx_exchange_floor", "yalps_dense_lp_f64", "yalps_dense_lp_rows_f64",
    "yalps_round_to_precision", "yalps_ctx_create_on_stream", "yalps_tableau_set_shard", "yalps_shard_slot_doubles",
    "yalps_shard_begin", "yalps_shard_select", "yalps_shard_apply", "yalps_shard_poll", "yalps_tableau_info",
    "yalps_tableau_assemble", "yalps_simplex_sparse_f64", "yalps_tableau_apply_cuts", "yalps_tableau_node_solve", "yalps_tableau_download_solution", "yalps_milp_f64", "yalps_batch_create", "yalps_batch_destroy", "yalps_batch_set_root", "yalps_batch_solve", "yalps_batch_download",
    "yalps_tableau_debug_stamps", "yalps_tableau_padding_check", "yalps_comm_unique_id", "yalps_comm_create", "yalps_comm_create_host", "yalps_comm_destroy",
    "yalps_comm_info", "yalps_shard_run",
)


class NativeError(RuntimeError):
    pass


# int32_t (*yalps_allgather_fn)(void *user, const double *send_host, double *recv_host, int64_t doubles_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        f64p, i32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
        L.yalps_last_error.restype = C.c_char_p
        L.yalps_device_count.restype = C.c_int32
        L.yalps_simplex_f64.restype = C.c_int32
        L.yalps_simplex_f64.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.c_double, C.c_double, C.c_int32, f64p]
        L.yalps_simplex_f64_ex.restype = C.c_int32
        L.yalps_simplex_f64_ex.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.c_double, C.c_double, C.c_int32,
                                           C.c_int32, f64p, C.POINTER(C.c_int64)]
        L.yalps_simplex_sparse_f64.restype = C.c_int32
        L.yalps_simplex_sparse_f64.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, C.c_double, C.c_double,
                                               C.c_int32, vp, vp, vp, f64p, C.POINTER(C.c_int64)]
        L.yalps_milp_f64.restype = C.c_int32
        L.yalps_milp_f64.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, C.c_double, C.c_double, C.c_double,
                                     C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_int32), f64p,
                                     vp, vp, vp, C.POINTER(C.c_int32), vp]
        L.yalps_tableau_download_solution.restype = C.c_int32
        L.yalps_tableau_download_solution.argtypes = [vp, vp, vp, vp]
        L.yalps_tableau_apply_cuts.restype = C.c_int32
        L.yalps_tableau_apply_cuts.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
        L.yalps_tableau_node_solve.restype = C.c_int32
        L.yalps_tableau_node_solve.argtypes = [vp, vp, C.c_int32, vp, vp, vp, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double), vp, vp, vp]
        L.yalps_tableau_assemble.restype = C.c_int32
        L.yalps_tableau_assemble.argtypes = [vp, C.c_int32, C.c_int64, vp, vp, vp]
        L.yalps_ctx_create.restype = C.c_int32
        L.yalps_ctx_create.argtypes = [C.c_int32, C.POINTER(vp)]
        L.yalps_ctx_create_on_stream.restype = C.c_int32
        L.yalps_ctx_create_on_stream.argtypes = [C.c_int32, vp, C.POINTER(vp)]
        L.yalps_tableau_set_shard.restype = C.c_int32
        L.yalps_tableau_set_shard.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, vp, vp]
        L.yalps_shard_slot_doubles.restype = C.c_int64
        L.yalps_shard_slot_doubles.argtypes = [vp]
        L.yalps_shard_begin.restype = C.c_int32
        L.yalps_shard_begin.argtypes = [vp, C.c_double, C.c_double, C.c_int32]
        L.yalps_shard_select.restype = C.c_int32
        L.yalps_shard_select.argtypes = [vp, vp]
        L.yalps_shard_apply.restype = C.c_int32
        L.yalps_shard_apply.argtypes = [vp, vp]
        L.yalps_shard_poll.restype = C.c_int32
        L.yalps_shard_poll.argtypes = [vp, C.POINTER(C.c_int32), f64p, C.POINTER(C.c_int64)]
        L.yalps_comm_unique_id.restype = C.c_int32
        L.yalps_comm_unique_id.argtypes = [vp]
        L.yalps_comm_create.restype = C.c_int32
        L.yalps_comm_create.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.yalps_comm_create_host.restype = C.c_int32
        L.yalps_comm_create_host.argtypes = [vp, ALLGATHER_FN, vp, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.yalps_comm_destroy.restype = None
        L.yalps_comm_destroy.argtypes = [vp]
        L.yalps_comm_info.restype = C.c_int32
        L.yalps_comm_info.argtypes = [vp, C.c_char_p, C.c_int32]
        L.yalps_shard_run.restype = C.c_int32
        L.yalps_shard_run.argtypes = [vp, vp, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32), f64p,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        L.yalps_batch_create.restype = C.c_int32
        L.yalps_batch_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.yalps_batch_destroy.restype = None
        L.yalps_batch_destroy.argtypes = [vp]
        L.yalps_batch_set_root.restype = C.c_int32
        L.yalps_batch_set_root.argtypes = [vp, vp, vp, vp]
        L.yalps_batch_solve.restype = C.c_int32
        L.yalps_batch_solve.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp,
                                        C.POINTER(C.c_float)]
        L.yalps_batch_download.restype = C.c_int32
        L.yalps_batch_download.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.yalps_ctx_destroy.restype = None
        L.yalps_ctx_destroy.argtypes = [vp]
        L.yalps_tableau_create.restype = C.c_int32
        L.yalps_tableau_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.yalps_tableau_destroy.restype = None
        L.yalps_tableau_destroy.argtypes = [vp]
        L.yalps_tableau_upload.restype = C.c_int32
        L.yalps_tableau_upload.argtypes = [vp, vp, C.c_int32, vp, vp]
        L.yalps_tableau_download.restype = C.c_int32
        L.yalps_tableau_download.argtypes = [vp, vp, vp, vp]
        L.yalps_tableau_download_rhs.restype = C.c_int32
        L.yalps_tableau_download_rhs.argtypes = [vp, vp]
        L.yalps_tableau_copy.restype = C.c_int32
        L.yalps_tableau_copy.argtypes = [vp, vp]
        L.yalps_tableau_height.restype = C.c_int32
        L.yalps_tableau_height.argtypes = [vp]
        L.yalps_tableau_info.restype = C.c_int32
        L.yalps_tableau_info.argtypes = [vp, C.c_char_p, C.c_int32]
        L.yalps_tableau_debug_stamps.restype = C.c_int32
        L.yalps_tableau_debug_stamps.argtypes = [vp, vp, C.c_int32, C.c_int32]
        L.yalps_tableau_solve.restype = C.c_int32
        L.yalps_tableau_solve.argtypes = [vp, C.c_double, C.c_double, C.c_int32, f64p, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_float)]
        L.yalps_tableau_pivot.restype = C.c_int32
        L.yalps_tableau_pivot.argtypes = [vp, C.c_int32, C.c_int32]
        L.yalps_tableau_bench_sweep.restype = C.c_int32
        L.yalps_tableau_bench_sweep.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.yalps_dense_lp_f64.restype = None
        L.yalps_dense_lp_f64.argtypes = [C.c_int32, C.c_int32, C.c_double, vp]
        L.yalps_dense_lp_rows_f64.restype = None
        L.yalps_dense_lp_rows_f64.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, vp]
        L.yalps_round_to_precision.restype = C.c_double
        L.yalps_round_to_precision.argtypes = [C.c_double, C.c_double]
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise NativeError("yalps_hip error %d: %s" % (rc, lib().yalps_last_error().decode()))
    return rc


def _ptr(a, dtype):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous, (type(a), getattr(a, "dtype", None))
    return a.ctypes.data


def simplex_host(matrix, width, height, pos, var, precision=1e-8, max_pivots=8192.0, check_cycles=False,
                 copyback=COPYBACK_FULL):
    """The drop-in: in-place on host numpy arrays, like the reference's simplex(tableau, options).
    Returns (status, result, n_pivots)."""
    assert matrix.size >= width * height
    res, npiv = C.c_double(), C.c_int64()
    st = check(lib().yalps_simplex_f64_ex(_ptr(matrix, np.float64), width, height, _ptr(pos, np.int32),
                                          _ptr(var, np.int32), precision, float(max_pivots), int(bool(check_cycles)),
                                          copyback, C.byref(res), C.byref(npiv)))
    return STATUS[st], res.value, npiv.value


def simplex_sparse(width, height, row, col, val, precision=1e-8, max_pivots=8192.0, check_cycles=False):
    """Initial tableau given by its written cells (sorted by (row, col)); assembled and solved in HBM.
    Returns (status, result, n_pivots, col0, positionOfVariable, variableAtPosition)."""
    assert row.size == col.size == val.size
    col0 = np.empty(height, np.float64)
    pos, var = np.empty(width + height, np.int32), np.empty(width + height, np.int32)
    res, npiv = C.c_double(), C.c_int64()
    st = check(lib().yalps_simplex_sparse_f64(width, height, row.size, _ptr(row, np.int32), _ptr(col, np.int32),
                                              _ptr(val, np.float64), precision, float(max_pivots),
                                              int(bool(check_cycles)), col0.ctypes.data, pos.ctypes.data,
                                              var.ctypes.data, C.byref(res), C.byref(npiv)))
    return STATUS[st], res.value, npiv.value, col0, pos, var


def milp(matrix, width, height, pos, var, integers, sign, precision=1e-8, max_pivots=8192.0, check_cycles=False,
         tolerance=0.0, timeout=float("inf"), max_iterations=32768.0, node_batch=0):
    """The whole branch and cut in one native call (yalps_milp_f64).  Returns (status, result, height, col0, pos, var,
    stats) -- what solution() reads of the best tableau."""
    ints = np.ascontiguousarray(integers, np.int32)
    extra = 2 * ints.size
    col0 = np.empty(height + extra, np.float64)
    opos, ovar = np.empty(width + height + extra, np.int32), np.empty(width + height + extra, np.int32)
    st, h, res, stats = C.c_int32(), C.c_int32(), C.c_double(), np.zeros(3, np.int64)
    check(lib().yalps_milp_f64(_ptr(matrix, np.float64), width, height, _ptr(pos, np.int32), _ptr(var, np.int32),
                               ints.ctypes.data, ints.size, float(sign), precision, float(max_pivots), int(bool(check_cycles)),
                               float(tolerance), float(timeout), float(max_iterations), int(node_batch), C.byref(st),
                               C.byref(res), col0.ctypes.data, opos.ctypes.data, ovar.ctypes.data, C.byref(h),
                               stats.ctypes.data))
    n = h.value
    return (STATUS[st.value], res.value, n, col0[:n].copy(), opos[:width + n].copy(), ovar[:width + n].copy(),
            {"nodes_used": int(stats[0]), "nodes_evaluated": int(stats[1]), "batches": int(stats[2])})


class Context:
    def __init__(self, device=0, stream=None):
        """stream: an existing HIP stream handle (int) to enqueue on, e.g.
        torch.cuda.current_stream().cuda_stream; None = a private stream."""
        self.handle = C.c_void_p()
        if stream is None:
            check(lib().yalps_ctx_create(device, C.byref(self.handle)))
        else:
            check(lib().yalps_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib().yalps_ctx_destroy(self.handle)
            self.handle = C.c_void_p()


def exchange_floor(ctx, workgroups=256, lanes=512, units=2, epochs=4000, variant=3):
    """us per round of the resident kernels' bare exchange on this chip (yalps_ctx_exchange_floor)."""
    us = C.c_float()
    check(lib().yalps_ctx_exchange_floor(ctx.handle, workgroups, lanes, units, epochs, variant, C.byref(us)))
    return us.value


class DeviceTableau:
    """A tableau resident in HBM."""

    def __init__(self, ctx, width, height_capacity):
        self.ctx, self.width, self.capacity = ctx, width, height_capacity
        self.handle = C.c_void_p()
        check(lib().yalps_tableau_create(ctx.handle, width, height_capacity, C.byref(self.handle)))

    @property
    def height(self):
        return lib().yalps_tableau_height(self.handle)

    def upload(self, matrix, height, pos, var):
        assert matrix.size >= self.width * height and pos.size >= self.width + height
        check(lib().yalps_tableau_upload(self.handle, _ptr(matrix, np.float64), height, _ptr(pos, np.int32),
                                         _ptr(var, np.int32)))

    def assemble(self, height, row, col, val):
        """Initial tableau from its written cells, sorted by (row, col) (yalps_tableau_assemble)."""
        assert row.size == col.size == val.size
        check(lib().yalps_tableau_assemble(self.handle, height, row.size, _ptr(row, np.int32), _ptr(col, np.int32),
                                           _ptr(val, np.float64)))

    def download(self, matrix=True, perms=True, perm_len=None):
        h, w = self.height, self.width
        n = perm_len if perm_len is not None else w + h
        m = np.empty(h * w, np.float64) if matrix else None
        pos = np.empty(n, np.int32) if perms else None
        var = np.empty(n, np.int32) if perms else None
        check(lib().yalps_tableau_download(self.handle, _ptr(m, np.float64), _ptr(pos, np.int32), _ptr(var, np.int32)))
        return m, pos, var

    def download_rhs(self):
        col0 = np.empty(self.height, np.float64)
        check(lib().yalps_tableau_download_rhs(self.handle, col0.ctypes.data))
        return col0

    def download_solution(self, perm_len=None):
        """(col0, positionOfVariable, variableAtPosition) with one wait (yalps_tableau_download_solution)."""
        n = perm_len if perm_len is not None else self.width + self.height
        col0, pos, var = np.empty(self.height, np.float64), np.empty(n, np.int32), np.empty(n, np.int32)
        check(lib().yalps_tableau_download_solution(self.handle, col0.ctypes.data, pos.ctypes.data, var.ctypes.data))
        return col0, pos, var

    def copy_from(self, other):
        check(lib().yalps_tableau_copy(self.handle, other.handle))

    def apply_cuts(self, root, cuts):
        """self = root's tableau + one row per cut (sign, variable, value), all on the device (yalps_tableau_apply_cuts)."""
        sign = np.array([c[0] for c in cuts] or [0], np.int32)
        var = np.array([c[1] for c in cuts] or [0], np.int32)
        val = np.array([c[2] for c in cuts] or [0.0], np.float64)
        check(lib().yalps_tableau_apply_cuts(self.handle, root.handle, len(cuts), sign.ctypes.data, var.ctypes.data,
                                             val.ctypes.data))

    def node_solve(self, root, cuts, precision=1e-8, max_pivots=8192.0, check_cycles=False):
        """applyCuts + simplex + what solution() reads, one native call (yalps_tableau_node_solve: three launches and one wait where the
        node takes the resident kernel).  Returns (status, result, height, col0, pos, var);
        the three arrays are meaningful for an optimal node."""
        sign = np.array([c[0] for c in cuts] or [0], np.int32)
        var = np.array([c[1] for c in cuts] or [0], np.int32)
        val = np.array([c[2] for c in cuts] or [0.0], np.float64)
        h = root.height + len(cuts)
        col0, p, v = np.empty(h), np.empty(self.width + h, np.int32), np.empty(self.width + h, np.int32)
        res = C.c_double()
        st = check(lib().yalps_tableau_node_solve(self.handle, root.handle, len(cuts), sign.ctypes.data, var.ctypes.data, val.ctypes.data,
                                                  precision, float(max_pivots), int(bool(check_cycles)), C.byref(res),
                                                  col0.ctypes.data, p.ctypes.data, v.ctypes.data))
        return STATUS[st], res.value, h, col0, p, v

    def solve(self, precision=1e-8, max_pivots=8192.0, check_cycles=False, timing=True):
        """Returns (status, result, n_pivots, gpu_ms); timing=False skips the HIP events (gpu_ms = 0)."""
        res, npiv, ms = C.c_double(), C.c_int64(), C.c_float()
        st = check(lib().yalps_tableau_solve(self.handle, precision, float(max_pivots), int(bool(check_cycles)),
                                             C.byref(res), C.byref(npiv), C.byref(ms) if timing else None))
        return STATUS[st], res.value, npiv.value, ms.value

    def info(self):
        buf = C.create_string_buffer(512)
        check(lib().yalps_tableau_info(self.handle, buf, 512))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(" ") if "=" in kv)

    def debug_stamps(self, reset=True):
        """Diagnostic build only: (workgroups, 24) uint64 stage sums of the persistent launches (yalps_tableau_debug_stamps)."""
        out = np.zeros(1024 * 24, np.uint64)
        n = check(lib().yalps_tableau_debug_stamps(self.handle, out.ctypes.data, out.size, int(bool(reset))))
        return out[:n].reshape(-1, 24)

    def padding_check(self):
        """(non-finite, non-zero) doubles in the row padding of the current device buffer (yalps_tableau_padding_check)."""
        bad, nz = C.c_int64(), C.c_int64()
        check(lib().yalps_tableau_padding_check(self.handle, C.byref(bad), C.byref(nz)))
        return bad.value, nz.value

    def pivot(self, row, col):
        check(lib().yalps_tableau_pivot(self.handle, row, col))

    def bench_sweep(self, row, col, launches):
        us = C.c_float()
        check(lib().yalps_tableau_bench_sweep(self.handle, row, col, launches, C.byref(us)))
        return us.value

    # ---- row-sharded solve steps (yalps_amd/sharded.py drives them) ----
    def set_shard(self, rank, nranks, bounds, global_height, pos, var):
        b = np.ascontiguousarray(bounds, np.int32)
        assert b.size == nranks + 1 and pos.size == self.width + global_height
        check(lib().yalps_tableau_set_shard(self.handle, rank, nranks, b.ctypes.data, global_height,
                                            _ptr(pos, np.int32), _ptr(var, np.int32)))

    def shard_slot_doubles(self):
        return int(lib().yalps_shard_slot_doubles(self.handle))

    def shard_begin(self, precision, max_pivots, check_cycles=False):
        check(lib().yalps_shard_begin(self.handle, precision, float(max_pivots), int(bool(check_cycles))))

    def shard_select(self, send_ptr):
        check(lib().yalps_shard_select(self.handle, C.c_void_p(send_ptr)))

    def shard_apply(self, gathered_ptr):
        check(lib().yalps_shard_apply(self.handle, C.c_void_p(gathered_ptr)))

    def shard_run(self, comm, precision=1e-8, max_pivots=8192.0, check_every=64, check_cycles=False):
        """The whole row-sharded solve natively (yalps_shard_run): no Python between two pivots.
        Returns (status code, result, n_pivots, gpu_ms)."""
        st, res, npiv, ms = C.c_int32(), C.c_double(), C.c_int64(), C.c_float()
        check(lib().yalps_shard_run(self.handle, comm.handle, precision, float(max_pivots), int(bool(check_cycles)), int(check_every), C.byref(st),
                                    C.byref(res), C.byref(npiv), C.byref(ms)))
        return st.value, res.value, npiv.value, ms.value

    def shard_poll(self):
        st, res, npiv = C.c_int32(), C.c_double(), C.c_int64()
        check(lib().yalps_shard_poll(self.handle, C.byref(st), C.byref(res), C.byref(npiv)))
        return st.value, res.value, npiv.value

    def close(self):
        if self.handle:
            lib().yalps_tableau_destroy(self.handle)
            self.handle = C.c_void_p()


class Comm:
    """This rank's end of the sharded solve's exchange (yalps_comm): RCCL, or a host all-gather callback."""

    def __init__(self, handle, keep=None):
        self.handle, self._keep = handle, keep

    @staticmethod
    def unique_id():
        """128 bytes from ncclGetUniqueId: made on rank 0, handed to every rank by the host's own channel."""
        buf = C.create_string_buffer(128)
        check(lib().yalps_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, unique_id, rank, nranks):
        h = C.c_void_p()
        check(lib().yalps_comm_create(ctx.handle, C.c_char_p(bytes(unique_id)), rank, nranks, C.byref(h)))
        return cls(h)

    @classmethod
    def host(cls, ctx, allgather, rank, nranks):
        """allgather(send: float64[n]) -> float64[nranks * n] on host arrays (e.g. a gloo all-gather)."""
        def thunk(_user, send, recv, n):
            try:
                out = allgather(np.ctypeslib.as_array(send, shape=(n,)).copy())
                np.ctypeslib.as_array(recv, shape=(nranks * n,))[:] = out
                return 0
            except Exception:  # (never unwind through the C frames)
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLGATHER_FN(thunk)
        h = C.c_void_p()
        check(lib().yalps_comm_create_host(ctx.handle, cb, None, rank, nranks, C.byref(h)))
        return cls(h, keep=cb)

    def info(self):
        buf = C.create_string_buffer(256)
        check(lib().yalps_comm_info(self.handle, buf, 256))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(" ") if "=" in kv)

    def close(self):
        if self.handle:
            lib().yalps_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class NodeBatch:
    """Batched branch-and-cut node evaluation: the root's optimal tableau stays in HBM, every node
    (= a list of cuts (sign, variable, value)) gets its own workgroup (yalps_batch_*)."""

    def __init__(self, ctx, width, root_height, max_cuts, max_nodes):
        self.ctx, self.width, self.root_height = ctx, width, root_height
        self.max_cuts, self.max_nodes = max_cuts, max_nodes
        self.handle = C.c_void_p()
        check(lib().yalps_batch_create(ctx.handle, width, root_height, max_cuts, max_nodes, C.byref(self.handle)))

    def set_root(self, matrix, pos, var):
        check(lib().yalps_batch_set_root(self.handle, _ptr(matrix, np.float64), _ptr(pos, np.int32), _ptr(var, np.int32)))

    def solve(self, cut_lists, precision=1e-8, max_pivots=8192.0):
        """cut_lists: per node a sequence of (sign, variable, value).  Returns (status names,
        results, pivot counts, node heights, gpu_ms)."""
        count = len(cut_lists)
        off = np.zeros(count + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in cut_lists])
        flat = [c for cuts in cut_lists for c in cuts]
        sign = np.array([c[0] for c in flat] or [0], np.int32)
        var = np.array([c[1] for c in flat] or [0], np.int32)
        val = np.array([c[2] for c in flat] or [0.0], np.float64)
        st, res, piv, ms = np.empty(count, np.int32), np.empty(count, np.float64), np.empty(count, np.int64), C.c_float()
        check(lib().yalps_batch_solve(self.handle, count, off.ctypes.data, sign.ctypes.data, var.ctypes.data,
                                      val.ctypes.data, precision, float(max_pivots), st.ctypes.data, res.ctypes.data,
                                      piv.ctypes.data, C.byref(ms)))
        heights = self.root_height + np.diff(off)
        return [STATUS[k] for k in st], res, piv, heights, ms.value

    def download(self, node, height, matrix=False):
        w = self.width
        m = np.empty(height * w, np.float64) if matrix else None
        col0 = np.empty(height, np.float64)
        pos, var = np.empty(w + height, np.int32), np.empty(w + height, np.int32)
        check(lib().yalps_batch_download(self.handle, node, height, _ptr(m, np.float64), col0.ctypes.data,
                                         pos.ctypes.data, var.ctypes.data))
        return m, col0, pos, var

    def close(self):
        if self.handle:
            lib().yalps_batch_destroy(self.handle)
            self.handle = C.c_void_p()


def dense_lp(M, N, seed=42.0):
    """dense-LP(M,N,seed) of SURVEY.md 8(d) as a flat row-major (M+1)x(N+1) tableau."""
    m = np.zeros((M + 1) * (N + 1), np.float64)
    lib().yalps_dense_lp_f64(M, N, float(seed), m.ctypes.data)
    return m


def dense_lp_rows(M, N, seed, row_begin, row_end):
    """Rows [row_begin, row_end) of dense-LP(M,N,seed) (row 0 = objective row), flat row-major: one rank's share of a
    row-sharded tableau without the 8*(M+1)*(N+1) bytes of the whole."""
    row_end = min(row_end, M + 1)
    m = np.zeros(max(row_end - row_begin, 0) * (N + 1), np.float64)
    lib().yalps_dense_lp_rows_f64(M, N, float(seed), row_begin, row_end, m.ctypes.data)
    return m


def round_to_precision(x, precision):
    return lib().yalps_round_to_precision(x, precision)
