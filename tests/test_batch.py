"""Batched branch-and-cut node evaluation on the GPU (BASELINE config 4):
node-level bit parity with the oracle, and the batched driver against the one-node-at-a-time
driver and the reference test-suite's expected results."""
import numpy as np
import pytest

from tests import _cases as K
from tests import _golden as G
from yalps_amd import branch_and_cut as BC
from yalps_amd import model as M
from yalps_amd import solve as S

pytestmark = pytest.mark.gpu
INTEGER_CASES = [n for n in K.names() if K.load(n)["model"].get("integers") or K.load(n)["model"].get("binaries")]


@pytest.fixture(scope="module")
def nat():
    from yalps_amd import _native
    assert _native.lib().yalps_device_count() >= 1
    return _native


def _collect_nodes(oracle, tabmod, init_result, options, limit):
    """Runs the sequential driver with the oracle and records the cut list of every node it evaluates."""
    from tests.test_host_model import oracle_backend
    base = oracle_backend(oracle)
    seen = []

    def spy(tableau, opt):
        extra = tableau.height - tabmod.tableau.height
        seen.append(extra)
        return base(tableau, opt)

    nodes = []
    orig = BC.apply_cuts

    def record(tableau, buf, cuts):
        if len(nodes) < limit:
            nodes.append(tuple(cuts))
        return orig(tableau, buf, cuts)

    BC.apply_cuts = record
    try:
        BC.branch_and_cut(spy, tabmod, init_result, options)
    finally:
        BC.apply_cuts = orig
    return nodes


@pytest.mark.parametrize("lds", [True, False], ids=["lds", "hbm"])
@pytest.mark.parametrize("name", ["Knapsack 1", "Large Farm MIP", "Fancy Stock Cutting Problem", "Integer Sports Complex Problem"])
def test_batch_nodes_match_oracle(nat, oracle, name, lds, monkeypatch):
    """lds: the node tableaux fit in LDS and batch_kernel<.., true> solves them there; hbm: the same
    nodes with YALPS_HIP_NO_LDS=1 (read when the batch is created) in the HBM workspace."""
    from tests.test_host_model import oracle_backend
    if not lds:
        monkeypatch.setenv("YALPS_HIP_NO_LDS", "1")
    case = K.load(name)
    opt = case["options"]
    tabmod = M.tableau_model(case["model"])
    status, result = oracle_backend(oracle)(tabmod.tableau, opt)
    assert status == "optimal"
    nodes = _collect_nodes(oracle, tabmod, result, opt, 48)
    assert nodes
    t = tabmod.tableau
    ctx = nat.Context(0)
    batch = nat.NodeBatch(ctx, t.width, t.height, 2 * len(tabmod.integers), len(nodes))
    try:
        batch.set_root(t.matrix, t.position_of_variable, t.variable_at_position)
        st, res, piv, heights, _ = batch.solve(nodes, opt["precision"], opt["maxPivots"])
        buf = (np.zeros(t.matrix.size + 2 * len(tabmod.integers) * t.width), np.zeros(t.width + t.height + 2 * len(tabmod.integers), np.int32),
               np.zeros(t.width + t.height + 2 * len(tabmod.integers), np.int32))
        for i, cuts in enumerate(nodes):
            cur = BC.apply_cuts(t, buf, cuts)
            m, pos, var = cur.matrix.copy(), cur.position_of_variable.copy(), cur.variable_at_position.copy()
            est, eres, epiv, _ = oracle.simplex(m, cur.width, cur.height, pos, var, precision=opt["precision"],
                                                max_pivots=opt["maxPivots"])
            assert (st[i], int(piv[i]), int(heights[i])) == (est, epiv, cur.height), (i, cuts)
            assert G.same_number(float(res[i]), eres)
            gm, col0, gpos, gvar = batch.download(i, cur.height, matrix=True)
            assert np.array_equal(gm.view(np.int64), m.view(np.int64)), (i, cuts)
            assert np.array_equal(col0.view(np.int64), m.reshape(cur.height, cur.width)[:, 0].view(np.int64))
            assert np.array_equal(gpos, pos) and np.array_equal(gvar, var)
    finally:
        batch.close()
        ctx.close()


@pytest.mark.parametrize("name", INTEGER_CASES)
def test_batched_solve_equals_sequential(nat, name):
    case = K.load(name)
    if case["options"].get("checkCycles"):
        pytest.skip("checkCycles uses the one-node-at-a-time path")
    stats = {}
    a = S.solve(case["model"], case["options"], node_batch=32, stats=stats, native=False)
    b = S.solve(case["model"], case["options"], device_nodes=False, native=False)
    assert a["status"] == b["status"] and G.same_number(a["result"], b["result"]) and a["variables"] == b["variables"]
    assert K.valid_solution_and_status(a, case["expected"], case["model"], case["options"])
    if "nodes_used" in stats:  # (roots above 4 MB take the device-resident one-node-at-a-time path instead)
        assert stats["nodes_used"] <= stats["nodes_evaluated"]


# ---- one node at a time with the root resident in HBM (yalps_tableau_apply_cuts) -------------------
@pytest.mark.parametrize("name", ["Knapsack 1", "Large Farm MIP", "Monster 2"])
def test_device_apply_cuts_matches_host(nat, oracle, name):
    """applyCuts on the device (rows built from the root's HBM copy) == the host restatement of
    src/branchAndCut.ts:22-61, bit for bit: matrix, RHS, both permutations."""
    from tests.test_host_model import oracle_backend
    case = K.load(name)
    tabmod = M.tableau_model(case["model"])
    t = tabmod.tableau
    status, result = oracle_backend(oracle)(t, case["options"])
    assert status == "optimal"
    nodes = _collect_nodes(oracle, tabmod, result, case["options"], 12)
    extra = 2 * len(tabmod.integers)
    ctx = nat.Context(0)
    root, node = nat.DeviceTableau(ctx, t.width, t.height), nat.DeviceTableau(ctx, t.width, t.height + extra)
    buf = (np.zeros(t.matrix.size + extra * t.width), np.zeros(t.width + t.height + extra, np.int32),
           np.zeros(t.width + t.height + extra, np.int32))
    try:
        root.upload(t.matrix, t.height, t.position_of_variable, t.variable_at_position)
        for cuts in nodes:
            cur = BC.apply_cuts(t, buf, cuts)
            node.apply_cuts(root, cuts)
            gm, gpos, gvar = node.download()
            assert node.height == cur.height
            assert np.array_equal(gm.view(np.int64), cur.matrix.view(np.int64)), cuts
            assert np.array_equal(gpos, cur.position_of_variable) and np.array_equal(gvar, cur.variable_at_position)
    finally:
        node.close()
        root.close()
        ctx.close()


@pytest.mark.parametrize("name", ["Large Farm MIP", "Monster 2", "Vendor Selection"])
def test_node_solve_fused_equals_call_by_call_and_oracle(nat, oracle, name, monkeypatch):
    """yalps_tableau_node_solve (src/branchAndCut.ts:126-127 on one node: applyCuts + simplex + column 0 and both permutations
    back): the fused form (node_prepare_kernel, the resident kernel, node_finish_kernel -- nodes that take the resident kernel:
    Monster 2, Vendor Selection) against the same node call by call, and both against the oracle on the host-built node
    tableau -- status, result, column 0 and permutations bit for bit; every count of cuts twice (staging reuse)."""
    from tests.test_host_model import oracle_backend
    case = K.load(name)
    tabmod = M.tableau_model(case["model"])
    t = tabmod.tableau
    opt = {**S.default_options, **case["options"]}
    status, result = oracle_backend(oracle)(t, opt)
    assert status == "optimal"
    nodes = _collect_nodes(oracle, tabmod, result, opt, 14)
    nodes = nodes + nodes[::-1]  # (every count of cuts at least twice: capture, then replays)
    extra = 2 * len(tabmod.integers)
    ctx = nat.Context(0)
    root, node = nat.DeviceTableau(ctx, t.width, t.height), nat.DeviceTableau(ctx, t.width, t.height + extra)
    buf = (np.zeros(t.matrix.size + extra * t.width), np.zeros(t.width + t.height + extra, np.int32),
           np.zeros(t.width + t.height + extra, np.int32))
    try:
        root.upload(t.matrix, t.height, t.position_of_variable, t.variable_at_position)
        for cuts in nodes:
            cur = BC.apply_cuts(t, buf, cuts)
            mm, pp, vv = cur.matrix.copy(), cur.position_of_variable.copy(), cur.variable_at_position.copy()
            est, eres, _, _ = oracle.simplex(mm, cur.width, cur.height, pp, vv, precision=opt["precision"],
                                             max_pivots=opt["maxPivots"], check_cycles=opt["checkCycles"])
            got = {}
            for fused in ("1", "0"):
                monkeypatch.setenv("YALPS_HIP_NODE_FUSED", fused)
                st, res, hgt, col0, pos, var = node.node_solve(root, cuts, opt["precision"], opt["maxPivots"], opt["checkCycles"])
                assert (st, hgt) == (est, cur.height) and G.same_number(res, eres), (cuts, fused)
                if st == "optimal":
                    assert np.array_equal(col0.view(np.int64), mm.reshape(cur.height, cur.width)[:, 0].copy().view(np.int64)), (cuts, fused)
                    assert np.array_equal(pos, pp) and np.array_equal(var, vv), (cuts, fused)
                got[fused] = (node.info()["last_path"], int(node.info()["node_fused_runs"]))
            assert got["1"][0] == got["0"][0]
        if name != "Large Farm MIP":  # (its nodes fit the LDS of one CU: small_kernel, call by call)
            assert got["1"][1] == len(nodes), got
    finally:
        node.close()
        root.close()
        ctx.close()


@pytest.mark.parametrize("name", INTEGER_CASES)
def test_device_nodes_solve_equals_sequential(nat, name):
    """The whole MILP with root and nodes resident in HBM (forced for every integer case, whatever its size)
    against the reference's flow through the host-array drop-in call."""
    case = K.load(name)
    opt = dict(S.default_options)
    opt.update(case["options"])
    stats = {}
    a = S._milp_on_device(M.tableau_model(case["model"], sparse=True), opt, stats)
    b = S.solve(case["model"], case["options"], device_nodes=False, native=False)
    assert a["status"] == b["status"] and G.same_number(a["result"], b["result"]) and a["variables"] == b["variables"]
    assert K.valid_solution_and_status(a, case["expected"], case["model"], case["options"])


# ---- the whole branch and cut in one native call (yalps_milp_f64) ----------------------------------
@pytest.mark.parametrize("node_batch", [0, 32], ids=["one-at-a-time", "batches-of-32"])
@pytest.mark.parametrize("name", INTEGER_CASES)
def test_native_branch_and_cut_equals_reference_flow(nat, name, node_batch):
    """yalps_milp_f64 (native heap with heapq / heap.js sift rules, nodes on the GPU) against the reference's flow
    restated in Python (one drop-in simplex() call per node): same status, objective and variables; with batches
    the same again (speculative evaluation commits in pop order)."""
    case = K.load(name)
    stats = {}
    a = S.solve(case["model"], case["options"], node_batch=node_batch, stats=stats, native=True)
    b = S.solve(case["model"], case["options"], device_nodes=False, native=False)
    assert a["status"] == b["status"] and G.same_number(a["result"], b["result"]) and a["variables"] == b["variables"]
    assert K.valid_solution_and_status(a, case["expected"], case["model"], case["options"])
    assert stats["nodes_used"] <= stats["nodes_evaluated"] or stats["nodes_evaluated"] == 0


def test_native_branch_and_cut_timeout_and_iteration_limit(nat):
    case = K.load("Knapsack 1")
    sol = S.solve(case["model"], {**case["options"], "timeout": 0})
    assert sol["status"] == "timedout"  # tests/solver.ts:126-135
    sol = S.solve(case["model"], {**case["options"], "maxIterations": 1})
    ref = S.solve(case["model"], {**case["options"], "maxIterations": 1}, native=False, device_nodes=False)
    assert sol["status"] == ref["status"] and G.same_number(sol["result"], ref["result"]) and sol["variables"] == ref["variables"]


def test_native_branch_and_cut_on_random_milps(nat):
    """Seeded random small MILPs (knapsack-like rows, integer and binary variables, both directions, a tolerance now
    and then): the native driver, one node at a time and in batches, against the Python restatement of the reference
    flow -- same status, objective and variables (so the same queue order and the same node LPs)."""
    rng = np.random.default_rng(77)
    seen = set()
    for case in range(60):
        nv, nc = int(rng.integers(2, 9)), int(rng.integers(1, 6))
        variables = {}
        for v in range(nv):
            coefs = {"c%d" % c: float(rng.integers(0, 9)) for c in range(nc) if rng.random() < 0.8}
            coefs["obj"] = float(rng.integers(1, 20))
            variables["x%d" % v] = coefs
        constraints = {"c%d" % c: ({"max": float(rng.integers(5, 40))} if rng.random() < 0.8 else
                                   {"min": float(rng.integers(1, 5)), "max": float(rng.integers(20, 60))}) for c in range(nc)}
        keys = list(variables)
        ints = [k for k in keys if rng.random() < 0.7]
        bins = [k for k in keys if k not in ints and rng.random() < 0.5]
        model = {"direction": "maximize" if rng.random() < 0.7 else "minimize", "objective": "obj", "constraints": constraints,
                 "variables": variables, "integers": ints, "binaries": bins}
        options = {"tolerance": float(rng.choice([0.0, 0.0, 0.05])), "maxIterations": int(rng.choice([32768, 32768, 6]))}
        ref = S.solve(model, options, native=False, device_nodes=False)
        seen.add(ref["status"])
        for nb in (0, 16):
            got = S.solve(model, options, node_batch=nb, native=True)
            assert got["status"] == ref["status"] and G.same_number(got["result"], ref["result"]), (case, nb, got, ref)
            assert got["variables"] == ref["variables"], (case, nb)
    assert "optimal" in seen and len(seen) >= 2, seen
