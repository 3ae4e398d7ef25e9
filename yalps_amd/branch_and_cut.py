"""Best-first branch and bound over variable-bound cuts: host-side caller of the hot path,
restating /root/reference/src/branchAndCut.ts:22-176.  Every node re-solves `root optimal
tableau + cut rows` with simplex() (:126-127).

Queue: the reference uses npm `heap` 0.2.7 (package.json:157), which is a port of Python's
heapq; heapq with an eval-only ordering therefore pops ties in the same order.
"""
import heapq
import math
import time

import numpy as np

from .model import Tableau, TableauModel


class _Branch:
    __slots__ = ("eval", "cuts")

    def __init__(self, ev, cuts):
        self.eval, self.cuts = ev, cuts

    def __lt__(self, other):  # comparator (x, y) => x[0] - y[0], :100
        return self.eval < other.eval


def _js_round(x):
    f = math.floor(x)
    return f + 1.0 if x - f >= 0.5 else float(f)


def apply_cuts(tableau, buf, cuts):
    """:22-61  new tableau = root tableau + one row per cut (sign, variable, value)."""
    matrix, pos, var = buf
    width, height = tableau.width, tableau.height
    n = tableau.matrix.size
    matrix[:n] = tableau.matrix
    for i, (sign, variable, value) in enumerate(cuts):
        r = (height + i) * width
        p = int(tableau.position_of_variable[variable])
        if p < width:
            matrix[r] = sign * value
            matrix[r + 1:r + width] = 0.0
            matrix[r + p] = sign
        else:
            row = (p - width) * width
            matrix[r] = sign * (value - matrix[row])
            matrix[r + 1:r + width] = -sign * matrix[row + 1:row + width]
    length = width + height + len(cuts)
    pos[:width + height] = tableau.position_of_variable
    var[:width + height] = tableau.variable_at_position
    ext = np.arange(width + height, length, dtype=np.int32)
    pos[width + height:length] = ext
    var[width + height:length] = ext
    return Tableau(matrix[:n + width * len(cuts)], width, height + len(cuts), pos[:length], var[:length])


def most_fractional_var(tableau, int_vars):
    """:64-85 (vectorised: the first variable with the strictly largest fractional part)"""
    if not len(int_vars):
        return 0, 0.0, 0.0
    ints = np.asarray(int_vars, np.int64)
    rows = tableau.position_of_variable[ints].astype(np.int64) - tableau.width
    basic = rows >= 0
    if not basic.any():
        return 0, 0.0, 0.0
    col0 = tableau.col0 if tableau.col0 is not None else tableau.matrix[::tableau.width]
    vals = col0[rows[basic]]
    f = np.floor(vals)
    with np.errstate(invalid="ignore"):
        frac = np.abs(vals - np.where(vals - f >= 0.5, f + 1.0, f))  # |val - Math.round(val)|
    frac = np.where(np.isnan(frac), -1.0, frac)  # (NaN never wins a `>` comparison)
    k = int(np.argmax(frac))  # first maximum = the reference's strict `>` scan
    if not frac[k] > 0.0:
        return 0, 0.0, 0.0
    return int(ints[basic][k]), float(vals[k]), float(frac[k])


def branch_and_cut(simplex, tabmod, init_result, options):
    """:89-176.  Returns (TableauModel of the best tableau, status, result)."""
    tableau, sign, integers = tabmod.tableau, tabmod.sign, tabmod.integers
    precision, max_iterations = options["precision"], options["maxIterations"]
    tolerance, timeout = options["tolerance"], options["timeout"]
    init_variable, init_value, init_frac = most_fractional_var(tableau, integers)
    if init_frac <= precision:
        return tabmod, "optimal", init_result

    branches = []
    heapq.heappush(branches, _Branch(init_result, [(-1, init_variable, float(math.ceil(init_value)))]))
    heapq.heappush(branches, _Branch(init_result, [(1, init_variable, float(math.floor(init_value)))]))

    max_extra_rows = len(integers) * 2
    matrix_length = tableau.matrix.size + max_extra_rows * tableau.width
    pos_var_length = tableau.position_of_variable.size + max_extra_rows

    def new_buffer():
        return (np.zeros(matrix_length, np.float64), np.zeros(pos_var_length, np.int32),
                np.zeros(pos_var_length, np.int32))

    candidate, solution_buf = new_buffer(), new_buffer()
    optimal_threshold = init_result * (1.0 - sign * tolerance)
    now = lambda: time.time() * 1000.0  # noqa: E731  Date.now()
    stop_time = timeout + now()
    timedout = now() >= stop_time
    solution_found = False
    best_eval = math.inf
    best_tableau = tableau
    it = 0
    while it < max_iterations and branches and best_eval >= optimal_threshold and not timedout:
        br = heapq.heappop(branches)
        relaxed_eval, cuts = br.eval, br.cuts
        if relaxed_eval > best_eval:
            break
        current = apply_cuts(tableau, candidate, cuts)
        status, result = simplex(current, options)
        if status == "optimal" and result < best_eval:
            variable, value, frac = most_fractional_var(current, integers)
            if frac <= precision:
                solution_found = True
                best_eval = result
                best_tableau = current
                candidate, solution_buf = solution_buf, candidate
            else:
                cuts_upper, cuts_lower = [], []
                for cut in cuts:
                    direction, v = cut[0], cut[1]
                    if v == variable:
                        (cuts_lower if direction < 0 else cuts_upper).append(cut)
                    else:
                        cuts_upper.append(cut)
                        cuts_lower.append(cut)
                cuts_lower.append((1, variable, float(math.floor(value))))
                cuts_upper.append((-1, variable, float(math.ceil(value))))
                heapq.heappush(branches, _Branch(result, cuts_upper))
                heapq.heappush(branches, _Branch(result, cuts_lower))
        timedout = now() >= stop_time
        it += 1

    unfinished = (timedout or it >= max_iterations) and bool(branches) and best_eval >= optimal_threshold
    status = "timedout" if unfinished else ("infeasible" if not solution_found else "optimal")
    return (TableauModel(best_tableau, sign, tabmod.variables, integers), status,
            best_eval if solution_found else math.nan)


def branch_and_cut_batched(tabmod, init_result, options, node_batch, stats=None):
    """branchAndCut (:89-176) with the node LPs evaluated on the GPU in batches (yalps_batch_*):
    whenever the popped node has no result yet, it and the next-best `node_batch - 1` frontier nodes
    are solved together (one workgroup per node, root resident, cuts applied on the device).  A
    node's LP depends only on the root and its cuts, so evaluating it early changes nothing; nodes
    are consumed in exactly the reference's pop order.  Returns what branch_and_cut returns."""
    from . import _native
    tableau, sign, integers = tabmod.tableau, tabmod.sign, tabmod.integers
    precision, max_iterations = options["precision"], options["maxIterations"]
    tolerance, timeout = options["tolerance"], options["timeout"]
    init_variable, init_value, init_frac = most_fractional_var(tableau, integers)
    if init_frac <= precision:
        return tabmod, "optimal", init_result

    branches = []
    heapq.heappush(branches, _Branch(init_result, ((-1, init_variable, float(math.ceil(init_value))),)))
    heapq.heappush(branches, _Branch(init_result, ((1, init_variable, float(math.floor(init_value))),)))

    max_extra_rows = len(integers) * 2
    ctx = _native.Context(0)
    batch = _native.NodeBatch(ctx, tableau.width, tableau.height, max_extra_rows, node_batch)
    batch.set_root(tableau.matrix, tableau.position_of_variable, tableau.variable_at_position)
    cache = {}
    if stats is not None:
        stats.update(batches=0, nodes_evaluated=0, nodes_used=0, pivots=0, gpu_ms=0.0)

    def evaluate(first):
        todo, seen = [first], {first}
        for br in heapq.nsmallest(node_batch - 1, branches):
            if br.cuts not in cache and br.cuts not in seen:
                todo.append(br.cuts)
                seen.add(br.cuts)
        st, res, piv, heights, ms = batch.solve(todo, precision, options["maxPivots"])
        for i, cuts in enumerate(todo):
            view = None
            if st[i] == "optimal":
                _, col0, pos, var = batch.download(i, int(heights[i]))
                view = Tableau(None, tableau.width, int(heights[i]), pos, var, col0)
            cache[cuts] = (st[i], float(res[i]), view)
        if stats is not None:
            stats["batches"] += 1
            stats["nodes_evaluated"] += len(todo)
            stats["pivots"] += int(piv.sum())
            stats["gpu_ms"] += ms

    optimal_threshold = init_result * (1.0 - sign * tolerance)
    now = lambda: time.time() * 1000.0  # noqa: E731
    stop_time = timeout + now()
    timedout = now() >= stop_time
    solution_found, best_eval, best_tableau, it = False, math.inf, tableau, 0
    try:
        while it < max_iterations and branches and best_eval >= optimal_threshold and not timedout:
            br = heapq.heappop(branches)
            relaxed_eval, cuts = br.eval, br.cuts
            if relaxed_eval > best_eval:
                break
            if cuts not in cache:
                evaluate(cuts)
            status, result, current = cache.pop(cuts)
            if stats is not None:
                stats["nodes_used"] += 1
            if status == "optimal" and result < best_eval:
                variable, value, frac = most_fractional_var(current, integers)
                if frac <= precision:
                    solution_found, best_eval, best_tableau = True, result, current
                else:
                    cuts_upper, cuts_lower = [], []
                    for cut in cuts:
                        direction, v = cut[0], cut[1]
                        if v == variable:
                            (cuts_lower if direction < 0 else cuts_upper).append(cut)
                        else:
                            cuts_upper.append(cut)
                            cuts_lower.append(cut)
                    cuts_lower.append((1, variable, float(math.floor(value))))
                    cuts_upper.append((-1, variable, float(math.ceil(value))))
                    heapq.heappush(branches, _Branch(result, tuple(cuts_upper)))
                    heapq.heappush(branches, _Branch(result, tuple(cuts_lower)))
            timedout = now() >= stop_time
            it += 1
    finally:
        batch.close()
        ctx.close()

    unfinished = (timedout or it >= max_iterations) and bool(branches) and best_eval >= optimal_threshold
    status = "timedout" if unfinished else ("infeasible" if not solution_found else "optimal")
    return (TableauModel(best_tableau, sign, tabmod.variables, integers), status,
            best_eval if solution_found else math.nan)


def branch_and_cut_device(tabmod, root, node, init_result, options, stats=None):
    """branchAndCut (:89-176), one node at a time like the reference, but with the root's optimal tableau
    resident in HBM (`root`, a DeviceTableau) and every node built next to it on the device
    (yalps_tableau_apply_cuts into `node`): per node only the cuts go up and column 0 + the permutations come
    back -- what most_fractional_var (:64-85) and solution() read.  `tabmod.tableau` is the root's view
    (col0 + permutations).  Returns what branch_and_cut returns."""
    tableau, sign, integers = tabmod.tableau, tabmod.sign, tabmod.integers
    precision, max_iterations = options["precision"], options["maxIterations"]
    tolerance, timeout = options["tolerance"], options["timeout"]
    init_variable, init_value, init_frac = most_fractional_var(tableau, integers)
    if init_frac <= precision:
        return tabmod, "optimal", init_result

    branches = []
    heapq.heappush(branches, _Branch(init_result, [(-1, init_variable, float(math.ceil(init_value)))]))
    heapq.heappush(branches, _Branch(init_result, [(1, init_variable, float(math.floor(init_value)))]))
    optimal_threshold = init_result * (1.0 - sign * tolerance)
    now = lambda: time.time() * 1000.0  # noqa: E731  Date.now()
    stop_time = timeout + now()
    timedout = now() >= stop_time
    solution_found, best_eval, best_tableau, it = False, math.inf, tableau, 0
    if stats is not None:
        stats.update(device_nodes=0)  # (no pivot count on this path: yalps_tableau_node_solve returns status, result, column 0 and the basis only)
    while it < max_iterations and branches and best_eval >= optimal_threshold and not timedout:
        br = heapq.heappop(branches)
        relaxed_eval, cuts = br.eval, br.cuts
        if relaxed_eval > best_eval:
            break
        # applyCuts + simplex + column 0 / permutations back: one native call (three launches, one wait)
        status, result, node_height, col0, pos, var = node.node_solve(root, cuts, precision, options["maxPivots"], options["checkCycles"])
        if stats is not None:
            stats["device_nodes"] += 1
        if status == "optimal" and result < best_eval:
            current = Tableau(None, tableau.width, node_height, pos, var, col0)
            variable, value, frac = most_fractional_var(current, integers)
            if frac <= precision:
                solution_found, best_eval, best_tableau = True, result, current
            else:
                cuts_upper, cuts_lower = [], []
                for cut in cuts:
                    direction, v = cut[0], cut[1]
                    if v == variable:
                        (cuts_lower if direction < 0 else cuts_upper).append(cut)
                    else:
                        cuts_upper.append(cut)
                        cuts_lower.append(cut)
                cuts_lower.append((1, variable, float(math.floor(value))))
                cuts_upper.append((-1, variable, float(math.ceil(value))))
                heapq.heappush(branches, _Branch(result, cuts_upper))
                heapq.heappush(branches, _Branch(result, cuts_lower))
        timedout = now() >= stop_time
        it += 1

    unfinished = (timedout or it >= max_iterations) and bool(branches) and best_eval >= optimal_threshold
    status = "timedout" if unfinished else ("infeasible" if not solution_found else "optimal")
    return (TableauModel(best_tableau, sign, tabmod.variables, integers), status,
            best_eval if solution_found else math.nan)
