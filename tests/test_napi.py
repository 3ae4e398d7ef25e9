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


SOLVE_CASES = ["Wiki 1", "Stigler Diet", "Monster Problem", "Infeasible 2", "Cycling Introductory Example", "Chvatal Cycling",
               "Knapsack 1", "Large Farm MIP", "Integer Sports Complex Problem", "Sudoku 4x4", "Monster 2"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOLVE_CASES)
def test_node_solve_returns_the_reference_solution(name):
    """yalps.js `solve(model, options)` under node -- model -> tableau on the JS host, simplex / branch and cut in the addon,
    Solution marshalled as src/YALPS.ts:8-50 does -- against the Python mirror of the same entry point and the reference
    test-suite's expected result for the model (status, objective, feasibility)."""
    from tests import _cases as K
    from yalps_amd import solve as S
    case = K.load(name)
    enc = lambda x: str(x) if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))) else x
    job = {"model": case["model"], "options": {k: enc(v) for k, v in case["options"].items()}}
    from yalps_amd import build
    build.build_hip()
    assert build.build_napi() is not None
    out = subprocess.run(["node", os.path.join(ROOT, "yalps_amd", "napi", "run_solve.js")], input=json.dumps(job),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert "error" not in res, res
    num = lambda x: float(x) if isinstance(x, str) else x
    got = {"status": res["status"], "result": num(res["result"]), "variables": [(k, num(v)) for k, v in res["variables"]]}
    ref = S.solve(case["model"], case["options"])
    assert got["status"] == ref["status"]
    assert got["result"] == ref["result"] or (got["result"] != got["result"] and ref["result"] != ref["result"])
    assert got["variables"] == ref["variables"]
    assert K.valid_solution_and_status(got, case["expected"], case["model"], case["options"])


@pytest.mark.gpu
def test_node_solve_readme_example():
    """README.md:63-79 of the reference through yalps.js (needs the GPU: skipped without one)."""
    from yalps_amd import _native
    if _native.lib().yalps_device_count() == 0:
        pytest.skip("no GPU")
    model = {"direction": "maximize", "objective": "profit",
             "constraints": {"wood": {"max": 300}, "labor": {"max": 110}, "storage": {"max": 400}},
             "variables": {"table": {"wood": 30, "labor": 5, "profit": 1200, "storage": 30},
                           "dresser": {"wood": 20, "labor": 10, "profit": 1600, "storage": 50}},
             "integers": ["table", "dresser"]}
    out = subprocess.run(["node", os.path.join(ROOT, "yalps_amd", "napi", "run_solve.js")], input=json.dumps({"model": model}),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"status": "optimal", "result": 14400, "variables": [["table", 8], ["dresser", 3]]}


def test_js_tableau_model_builds_the_reference_tableaux():
    """yalps.js's model -> tableau conversion (CPU only: the addon is stubbed out) on all 46 models of the reference
    test-suite: width, height, sign, integer columns and the matrix bytes equal the Python mirror's, which
    tests/test_host_model.py pins to the tableaux the reference's own tableauModel built (tests/golden)."""
    from tests import _cases as K
    from yalps_amd import model as M
    js = ("const path=require('path'), Module=require('module'); const orig=Module._load;"
          "Module._load=function(req,...a){ if(req.endsWith('yalps_napi.node')) return {}; return orig.call(this,req,...a)};"
          "const y=require(path.resolve('yalps_amd/napi/yalps.js'));"
          "const cases=JSON.parse(require('fs').readFileSync(0,'utf-8'));"
          "console.log(JSON.stringify(cases.map(c=>{const t=y.tableauModel(c,0); return {w:t.width,h:t.height,sign:t.sign,"
          "ints:t.integers,m:Buffer.from(t.matrix.buffer).toString('hex')}})));")
    names = K.names()
    models = [K.load(n)["model"] for n in names]
    out = subprocess.run(["node", "-e", js], input=json.dumps(models), capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    for name, mdl, r in zip(names, models, json.loads(out.stdout)):
        tm = M.tableau_model(mdl)
        m = np.frombuffer(bytes.fromhex(r["m"]), np.float64)
        assert (r["w"], r["h"], r["sign"], r["ints"]) == (tm.tableau.width, tm.tableau.height, tm.sign, tm.integers), name
        assert np.array_equal(m.view(np.int64), tm.tableau.matrix.view(np.int64)), name
