"""solve(model, options) -> Solution: the reference's public entry point
(/root/reference/src/YALPS.ts:8-92) with the dense-tableau simplex running on the MI355X.

Host side (model construction, branch and cut, result marshalling) mirrors the reference; the
hot path -- `simplex(tableau, options)`, src/simplex.ts:144 -- is libyalps_hip.so.  There is no
CPU fallback: without the HIP library and a gfx950 device `solve` raises.
"""
import math

from . import _native
from .branch_and_cut import branch_and_cut
from .model import tableau_model

# src/YALPS.ts:52-60
_DEFAULTS = {
    "precision": 1e-8,
    "checkCycles": False,
    "maxPivots": 8192,
    "tolerance": 0,
    "timeout": math.inf,
    "maxIterations": 32768,
    "includeZeroVariables": False,
}

default_options = dict(_DEFAULTS)  # a copy, like the reference's exported `defaultOptions` (:65)

NODE_BATCH_MAX_BYTES = 4 << 20  # root tableaux above this size are never batched (see _solve_with)
SPARSE_MIN_BYTES = 128 << 10  # below this the dense tableau goes through the single-workgroup path (small_kernel), which
#                               reads it in place from pinned host memory: nothing to save by shipping cells


def round_to_precision(num, precision):
    """src/util.ts:1-4 with JS Math.round (halves toward +infinity)."""
    def js_round(x):
        if x != x or math.isinf(x):
            return x
        f = math.floor(x)
        return f + 1.0 if x - f >= 0.5 else float(f)
    if precision == 0:
        rounding = math.inf
    else:
        rounding = js_round(1.0 / precision)
    v = (num + 2.220446049250313e-16) * rounding
    if math.isinf(rounding):
        return math.nan if (v != v or math.isinf(v)) else v / rounding
    return js_round(v) / rounding


def hip_simplex(tableau, options):
    """The drop-in for src/simplex.ts:144 `simplex(tableau, options)`: in place, through the C ABI
    (yalps_simplex_f64).  Returns (status, result).  A tableau built with sparse=True that has no
    dense matrix yet goes up as its written cells and only column 0 + the permutations come back
    (yalps_simplex_sparse_f64) -- all that solution() reads (src/YALPS.ts:18-19,32)."""
    if tableau.matrix is None:
        row, col, val = tableau.cells
        status, result, _, tableau.col0, tableau.position_of_variable, tableau.variable_at_position = \
            _native.simplex_sparse(tableau.width, tableau.height, row, col, val, precision=options["precision"],
                                   max_pivots=options["maxPivots"], check_cycles=options["checkCycles"])
        return status, result
    status, result, _ = _native.simplex_host(
        tableau.matrix, tableau.width, tableau.height, tableau.position_of_variable, tableau.variable_at_position,
        precision=options["precision"], max_pivots=options["maxPivots"], check_cycles=options["checkCycles"])
    return status, result


def solution(tabmod, status, result, options):
    """src/YALPS.ts:8-50"""
    tableau, sign, vars_ = tabmod.tableau, tabmod.sign, tabmod.variables
    precision = options["precision"]
    if status == "optimal" or (status == "timedout" and not math.isnan(result)):
        variables = []
        for i, (key, _) in enumerate(vars_):
            row = int(tableau.position_of_variable[i + 1]) - tableau.width
            value = tableau.rhs(row) if row >= 0 else 0.0
            if value > precision:
                variables.append((key, round_to_precision(value, precision)))
            elif options["includeZeroVariables"]:
                variables.append((key, 0.0))
        return {"status": status, "result": -sign * result, "variables": variables}
    if status == "unbounded":
        variable = int(tableau.variable_at_position[int(result)]) - 1
        return {"status": "unbounded", "result": sign * math.inf,
                "variables": [(vars_[variable][0], math.inf)] if 0 <= variable < len(vars_) else []}
    return {"status": status, "result": math.nan, "variables": []}  # infeasible | cycled | timedout w/o result


def _milp_on_device(tabmod, opt, stats=None):
    """A MILP whose root tableau is too big for the single-workgroup path: the root is assembled (sparse) or
    uploaded once, solved and KEPT in HBM; branch and cut builds every node next to it on the device
    (branch_and_cut_device).  No tableau ever comes back to the host."""
    from .branch_and_cut import branch_and_cut_device
    from .model import Tableau, TableauModel
    t = tabmod.tableau
    ctx = _native.Context(0)
    root = _native.DeviceTableau(ctx, t.width, t.height)
    node = None
    try:
        if t.matrix is None:
            root.assemble(t.height, *t.cells)
        else:
            root.upload(t.matrix, t.height, t.position_of_variable, t.variable_at_position)
        status, result, _, _ = root.solve(opt["precision"], opt["maxPivots"], opt["checkCycles"])
        col0, pos, var = root.download_solution()
        view = TableauModel(Tableau(None, t.width, t.height, pos, var, col0), tabmod.sign,
                            tabmod.variables, tabmod.integers)
        if status != "optimal":
            return solution(view, status, result, opt)
        node = _native.DeviceTableau(ctx, t.width, t.height + 2 * len(tabmod.integers))
        int_tabmod, int_status, int_result = branch_and_cut_device(view, root, node, result, opt, stats)
        return solution(int_tabmod, int_status, int_result, opt)
    finally:
        if node is not None:
            node.close()
        root.close()
        ctx.close()


def _milp_native(tabmod, opt, node_batch=0, stats=None):
    """A model with integers through yalps_milp_f64: root simplex and the whole branch and cut (the reference's
    queue order, every node LP on the GPU) in ONE native call; only what solution() reads comes back."""
    from .model import Tableau, TableauModel
    t = tabmod.tableau
    status, result, height, col0, pos, var, st = _native.milp(
        t.dense(), t.width, t.height, t.position_of_variable, t.variable_at_position, tabmod.integers, tabmod.sign,
        precision=opt["precision"], max_pivots=opt["maxPivots"], check_cycles=opt["checkCycles"], tolerance=opt["tolerance"],
        timeout=opt["timeout"], max_iterations=opt["maxIterations"], node_batch=node_batch)
    if stats is not None:
        stats.update(st)
    view = TableauModel(Tableau(None, t.width, height, pos, var, col0), tabmod.sign, tabmod.variables, tabmod.integers)
    return solution(view, status, result, opt)


def _solve_with(simplex, model, options=None, node_batch=0, stats=None, sparse=False, device_nodes=False, native=False):
    """src/YALPS.ts:73-92 with the simplex backend as a parameter (tests drive the host logic
    with the CPU oracle through this; the product binds the HIP backend below)."""
    tabmod = tableau_model(model, sparse=sparse)
    opt = dict(_DEFAULTS)
    if options:
        opt.update({k: v for k, v in options.items() if v is not None})
    nbytes = 8 * tabmod.tableau.width * tabmod.tableau.height
    if native and tabmod.integers:
        return _milp_native(tabmod, opt, node_batch, stats)
    if device_nodes and tabmod.integers and nbytes > SPARSE_MIN_BYTES and not (node_batch > 1 and nbytes <= NODE_BATCH_MAX_BYTES):
        return _milp_on_device(tabmod, opt, stats)
    if sparse and (tabmod.integers or nbytes <= SPARSE_MIN_BYTES):
        tabmod.tableau.dense()  # branch and cut reads the whole root matrix (src/branchAndCut.ts:28,38-41)
    status, result = simplex(tabmod.tableau, opt)
    if not tabmod.integers or status != "optimal":
        return solution(tabmod, status, result, opt)
    # one workgroup per node only pays while a node's tableau is small (it streams through one CU);
    # large roots (Vendor Selection: 23 MB) are better off on the whole-chip kernels, one node at a time
    small = 8 * tabmod.tableau.width * tabmod.tableau.height <= NODE_BATCH_MAX_BYTES
    if node_batch > 1 and not opt["checkCycles"] and small:
        from .branch_and_cut import branch_and_cut_batched
        int_tabmod, int_status, int_result = branch_and_cut_batched(tabmod, result, opt, node_batch, stats)
    else:
        int_tabmod, int_status, int_result = branch_and_cut(simplex, tabmod, result, opt)
    return solution(int_tabmod, int_status, int_result, opt)


def solve(model, options=None, node_batch=None, stats=None, sparse=True, device_nodes=True, native=True):
    """Runs the solver on `model` (see yalps_amd.model) with `options` (keys as in the reference's
    `Options`, src/types.ts:203-265).  Returns {"status", "result", "variables": [(key, value)]}.

    node_batch > 1: branch and cut evaluates that many frontier nodes per GPU batch (speculatively,
    best first; results are committed in the reference's pop order, so the outcome is the same as
    node_batch = 0, which re-solves one node at a time).  Default: 32 with the native driver
    (Large Farm MIP: 120 ms one node at a time, 28 ms in batches of 32), 0 with the Python drivers.

    sparse: a model without integer variables is shipped as the cells tableauModel writes and its
    tableau is assembled in HBM (same tableau, same pivots, 16 B per cell over PCIe instead of
    8*width*height); False = always the dense host tableau.

    device_nodes: a MILP whose root tableau exceeds the single-workgroup size keeps it in HBM and builds
    every branch-and-cut node there (yalps_tableau_apply_cuts); False = the reference's flow, every node
    through the host-array drop-in call.

    native: a model with integers is handed to yalps_milp_f64 -- root simplex and the whole branch and cut in
    one native call (same queue order, same node LPs); False = the Python drivers of branch_and_cut.py."""
    if node_batch is None:
        node_batch = 32 if native else 0
    return _solve_with(hip_simplex, model, options, node_batch, stats, sparse, device_nodes, native)
