"""The N-API shim (yalps_amd/napi): builds against node's headers, loads under node, exposes
simplex(tableau, options) -> [status, number] like the reference export (src/simplex.ts:144)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "yalps_amd", "napi", "run_simplex.js")
README = {"matrix": [0, 1200, 1600, 300, 30, 20, 110, 5, 10, 400, 30, 50], "width": 3, "height": 4, "viewOffset": 4}

pytestmark = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / node_api.h not available")


def run_node(job):
    from yalps_amd import build
    build.build_hip()
    assert build.build_napi() is not None
    out = subprocess.run(["node", DRIVER], input=json.dumps(job), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_addon_loads_and_fails_loudly_without_gpu():
    from yalps_amd import _native
    if _native.lib().yalps_device_count() > 0:
        pytest.skip("a GPU is present")
    res = run_node(README)
    assert "no HIP device" in res["error"]  # thrown as a JS Error; no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 2, 0), (40, 30, 7), (120, 200, 3)])
def test_addon_matches_oracle(oracle, shape):
    M, N, off = shape
    w, h = N + 1, M + 1
    m = oracle.dense_lp(M, N, 11) if M > 3 else np.array(README["matrix"], np.float64)
    if M <= 3:
        w, h = 3, 4
    res = run_node({"matrix": m.tolist(), "width": w, "height": h, "viewOffset": off,
                    "options": {"maxPivots": "Infinity"}})
    assert "error" not in res, res
    ref = m.copy()
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    status, result, _, _ = oracle.simplex(ref, w, h, pos, var, max_pivots=np.inf)
    assert res["status"] == status and float(res["result"]) == result
    assert np.array_equal(np.array(res["matrix"]).view(np.int64), ref.view(np.int64))  # JSON round-trips doubles exactly
    assert res["positionOfVariable"] == pos.tolist() and res["variableAtPosition"] == var.tolist()
    assert res["guardsIntact"]  # typed-array views honoured: nothing written outside them


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Knapsack 1", "Large Farm MIP", "Fancy Stock Cutting Problem"])
def test_addon_branch_and_cut_exports(oracle, name):
    """rootSolve / nodeSolve / rootFree under node: the root's optimal tableau stays in HBM, every node is built
    there from its cut list; root and nodes must equal the oracle on the host-side applyCuts, bit for bit."""
    from tests import _cases as K
    from tests.test_batch import _collect_nodes
    from tests.test_host_model import oracle_backend
    from yalps_amd import branch_and_cut as BC, model as M
    case = K.load(name)
    opt = case["options"]
    tabmod = M.tableau_model(case["model"])
    t = tabmod.tableau
    init = t.matrix.copy()
    status, result = oracle_backend(oracle)(t, opt)  # t now holds the root's optimal tableau
    assert status == "optimal"
    nodes = _collect_nodes(oracle, tabmod, result, opt, 10)
    extra = 2 * len(tabmod.integers)
    job = {"matrix": init.tolist(), "width": t.width, "height": t.height, "maxCuts": extra,
           "options": {"precision": opt["precision"], "maxPivots": opt["maxPivots"], "checkCycles": opt["checkCycles"]},
           "nodes": [[[int(s), int(v), float(x)] for s, v, x in cuts] for cuts in nodes]}
    out = subprocess.run(["node", os.path.join(ROOT, "yalps_amd", "napi", "run_nodes.js")], input=json.dumps(job),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert "error" not in res, res
    assert res["status"] == "optimal" and float(res["result"]) == result
    f64 = lambda h: np.frombuffer(bytes.fromhex(h), np.float64)  # (hex of the bytes: JSON would turn -0 into 0)
    assert np.array_equal(f64(res["col0"]).view(np.int64), t.matrix.reshape(t.height, t.width)[:, 0].view(np.int64))
    assert res["positionOfVariable"] == t.position_of_variable.tolist()
    buf = (np.zeros(t.matrix.size + extra * t.width), np.zeros(t.width + t.height + extra, np.int32),
           np.zeros(t.width + t.height + extra, np.int32))
    for cuts, got in zip(nodes, res["nodes"]):
        cur = BC.apply_cuts(t, buf, cuts)
        m, pos, var = cur.matrix.copy(), cur.position_of_variable.copy(), cur.variable_at_position.copy()
        est, eres, _, _ = oracle.simplex(m, cur.width, cur.height, pos, var, precision=opt["precision"],
                                         max_pivots=opt["maxPivots"], check_cycles=opt["checkCycles"])
        assert got["status"] == est and got["height"] == cur.height
        assert (eres != eres and got["result"] == "NaN") or float(got["result"]) == eres
        assert np.array_equal(f64(got["col0"]).view(np.int64), m.reshape(cur.height, cur.width)[:, 0].view(np.int64))
        assert got["positionOfVariable"] == pos.tolist() and got["variableAtPosition"] == var.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name,node_batch", [("Knapsack 1", 0), ("Large Farm MIP", 32), ("Integer Sports Complex Problem", 8)])
def test_addon_solve_integer(name, node_batch):
    """solveInteger under node = the whole solve() of a model with integers in one native call: the Solution built
    from what it returns equals the Python restatement of the reference flow."""
    from tests import _cases as K
    from yalps_amd import model as M, solve as S
    from yalps_amd.model import Tableau, TableauModel
    case = K.load(name)
    opt = dict(S.default_options)
    opt.update(case["options"])
    tabmod = M.tableau_model(case["model"])
    t = tabmod.tableau
    enc = lambda x: "Infinity" if x == float("inf") else x
    job = {"matrix": t.matrix.tolist(), "width": t.width, "height": t.height, "integers": tabmod.integers, "sign": tabmod.sign,
           "nodeBatch": node_batch, "options": {k: enc(opt[k]) for k in ("precision", "maxPivots", "checkCycles", "tolerance",
                                                                           "timeout", "maxIterations")}}
    out = subprocess.run(["node", os.path.join(ROOT, "yalps_amd", "napi", "run_milp.js")], input=json.dumps(job),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert "error" not in res, res
    view = TableauModel(Tableau(None, t.width, res["height"], np.array(res["positionOfVariable"], np.int32),
                                np.array(res["variableAtPosition"], np.int32),
                                np.frombuffer(bytes.fromhex(res["col0"]), np.float64)), tabmod.sign, tabmod.variables, tabmod.integers)
    result = float(res["result"]) if not isinstance(res["result"], str) else float("nan")
    got = S.solution(view, res["status"], result, opt)
    ref = S.solve(case["model"], case["options"], native=False, device_nodes=False)
    assert got["status"] == ref["status"] and (got["result"] == ref["result"] or (got["result"] != got["result"] and ref["result"] != ref["result"]))
    assert got["variables"] == ref["variables"]
    assert K.valid_solution_and_status(got, case["expected"], case["model"], case["options"])
