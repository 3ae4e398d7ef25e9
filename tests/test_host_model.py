"""Host-side logic on CPU: the tableau builder against the tableaux the reference itself built,
and solve()/branch-and-cut driven by the CPU oracle against the reference test-suite's expected
results (reference parity definition: tests/helpers/validate.ts, restated in tests/_cases.py)."""
import math

import numpy as np
import pytest

from tests import _cases as K
from tests import _golden as G
from yalps_amd import model as M
from yalps_amd import solve as S

CASE_RECORDS = {r["name"]: r for r in G.records("cases")}


def oracle_backend(oracle):
    def simplex(tableau, options):
        status, result, _, _ = oracle.simplex(tableau.matrix, tableau.width, tableau.height,
                                              tableau.position_of_variable, tableau.variable_at_position,
                                              precision=options["precision"], max_pivots=options["maxPivots"],
                                              check_cycles=options["checkCycles"])
        return status, result
    return simplex


@pytest.mark.parametrize("name", K.names())
def test_tableau_model_matches_reference(name):
    """Bit-identical initial tableau, sign and integer list for every reference test model."""
    rec = CASE_RECORDS[name]
    tm = M.tableau_model(K.load(name)["model"])
    t = tm.tableau
    assert (t.width, t.height) == (rec["width"], rec["height"])
    assert G.sha256(t.matrix) == rec["init_sha256"]
    assert tm.sign == rec["sign"] and tm.integers == rec["integers"]
    n = t.width + t.height
    assert np.array_equal(t.position_of_variable, np.arange(n)) and np.array_equal(t.variable_at_position, np.arange(n))


def test_empty_model_exact_layout():
    tm = M.tableau_model({"variables": {}, "constraints": {}})  # reference tests/tableau.ts:12-27
    t = tm.tableau
    assert (t.width, t.height, t.matrix.tolist()) == (1, 1, [0.0])
    assert t.position_of_variable.tolist() == [0, 1] and t.variable_at_position.tolist() == [0, 1]
    assert tm.sign == 1.0 and tm.variables == [] and tm.integers == []


@pytest.mark.parametrize("name", [n for n in K.names() if n not in K.LARGE])
def test_tableau_model_properties(name):
    """A few of the reference's metamorphic properties (tests/tableau.ts:49-67,104-133,223-242)."""
    model = K.load(name)["model"]
    base = M.tableau_model(model)
    w = base.tableau.width
    # no objective -> zero objective row
    noobj = M.tableau_model({**model, "objective": None})
    exp = base.tableau.matrix.copy()
    exp[:w] = 0.0
    assert np.array_equal(noobj.tableau.matrix, exp)
    # opposite direction -> objective row and sign negated
    flipped = M.tableau_model({**model, "direction": "maximize" if model.get("direction") == "minimize" else "minimize"})
    assert flipped.sign == -base.sign
    assert np.array_equal(flipped.tableau.matrix[:w], -base.tableau.matrix[:w])
    assert np.array_equal(flipped.tableau.matrix[w:], base.tableau.matrix[w:])
    # dict / list-of-pairs equivalence for constraints, variables and coefficients
    as_pairs = {**model, "constraints": M.entries(model["constraints"]),
                "variables": [(k, M.entries(v)) for k, v in M.entries(model["variables"])]}
    assert np.array_equal(M.tableau_model(as_pairs).tableau.matrix, base.tableau.matrix)
    # `equal` overrides min/max
    cons = {k: ({**c, "min": -1e9, "max": 1e9} if "equal" in c else c) for k, c in M.entries(model["constraints"])}
    assert np.array_equal(M.tableau_model({**model, "constraints": cons}).tableau.matrix, base.tableau.matrix)


def test_js_property_order_for_dict_models():
    # integer-like keys come first in ascending order, like Object.entries on a JS object
    assert [k for k, _ in M.entries({"b": 1, "10": 2, "2": 3, "a": 4, "01": 5})] == ["2", "10", "b", "a", "01"]


@pytest.mark.parametrize("name", K.names())
def test_solve_with_oracle_backend_matches_expected(oracle, name):
    """solve() host logic (builder + marshalling + branch and cut) is right: with the CPU oracle as
    the simplex it reproduces status/objective/feasibility of every reference test case."""
    case = K.load(name)
    sol = S._solve_with(oracle_backend(oracle), case["model"], case["options"])
    assert K.valid_solution_and_status(sol, case["expected"], case["model"], case["options"]), sol["status"]
    # variable order preserved (tests/solver.ts:27-47)
    keys = [k for k, _ in M.entries(case["model"]["variables"])]
    it = iter(keys)
    assert all(any(k == key for k in it) for key, _ in sol["variables"])


def test_readme_example(oracle):
    model = {"direction": "maximize", "objective": "profit",
             "constraints": {"wood": M.less_eq(300), "labor": M.less_eq(110), "storage": M.less_eq(400)},
             "variables": {"table": {"wood": 30, "labor": 5, "profit": 1200, "storage": 30},
                           "dresser": {"wood": 20, "labor": 10, "profit": 1600, "storage": 50}},
             "integers": ["table", "dresser"]}
    sol = S._solve_with(oracle_backend(oracle), model)
    assert sol == {"status": "optimal", "result": 14400.0, "variables": [("table", 8.0), ("dresser", 3.0)]}
    relaxed = S._solve_with(oracle_backend(oracle), {**model, "integers": None})
    assert relaxed["result"] == 14666.66666667
    assert relaxed["variables"] == [("table", 7.77777778), ("dresser", 3.33333333)]


def test_timeout_and_options(oracle):
    case = K.load("Knapsack 1")
    sol = S._solve_with(oracle_backend(oracle), case["model"], {**case["options"], "timeout": 0})
    assert sol["status"] == "timedout"  # tests/solver.ts:126-135
    sol = S._solve_with(oracle_backend(oracle), case["model"], {**case["options"], "includeZeroVariables": True})
    assert [k for k, _ in sol["variables"]] == [k for k, _ in M.entries(case["model"]["variables"])]
    assert S.default_options["maxPivots"] == 8192 and S.default_options["precision"] == 1e-8


def test_round_to_precision_matches_oracle(oracle):
    for x in (-14666.666666666668, 2.5e-9, -2.5e-9, 0.0, 1e300, -0.5, 0.49999999999999994, math.inf):
        for p in (1e-8, 1e-6, 1.0, 0.3):
            a, b = S.round_to_precision(x, p), oracle.round_to_precision(x, p)
            assert G.same_number(a, b), (x, p, a, b)
    assert math.isnan(S.round_to_precision(1.0, 0.0))


@pytest.mark.parametrize("name", K.names())
def test_sparse_tableau_model_same_cells(name):
    """tableau_model(sparse=True): the written cells, sorted by (row, col) without
    duplicates, rebuild the reference's initial tableau bit for bit."""
    rec = CASE_RECORDS[name]
    tm = M.tableau_model(K.load(name)["model"], sparse=True)
    t = tm.tableau
    assert t.matrix is None and (t.width, t.height) == (rec["width"], rec["height"])
    row, col, val = t.cells
    assert row.dtype == np.int32 and col.dtype == np.int32 and val.dtype == np.float64
    flat = row.astype(np.int64) * t.width + col
    assert np.all(np.diff(flat) > 0) and (flat.size == 0 or (flat[0] >= 0 and flat[-1] < t.width * t.height))
    assert G.sha256(t.dense()) == rec["init_sha256"]
    assert tm.sign == rec["sign"] and tm.integers == rec["integers"]


def test_sparse_tableau_model_last_duplicate_wins():
    """src/tableau.ts:101-115: a later coefficient for the same (variable, constraint) overwrites."""
    model = {"objective": "p", "constraints": [("a", {"max": 4}), ("a", {"min": 1}), ("b", {"max": 9})],
             "variables": [("x", [("a", 1.0), ("p", 2.0), ("a", 3.0)]), ("y", [("b", 5.0), ("b", 0.0), ("p", 7.0)])]}
    dense, sparse = M.tableau_model(model), M.tableau_model(model, sparse=True)
    assert np.array_equal(sparse.tableau.dense(), dense.tableau.matrix)
    assert dense.tableau.matrix.reshape(dense.tableau.height, -1).tolist() == \
        [[0, 2, 7], [4, 3, 0], [-1, -3, 0], [9, 0, 0]]
