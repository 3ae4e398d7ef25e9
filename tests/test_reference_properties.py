"""The reference test-suite's metamorphic properties, restated for the Python mirror of its host API
(/root/reference/tests/tableau.ts:49-360 for tableauModel, tests/solver.ts:27-135 for solve()).  They run on
CPU: tableau_model needs no device, solve() is driven with the CPU oracle as its simplex backend (the same host
code the GPU path runs behind; tests/test_hip_parity.py::test_solve_matches_reference_cases ties the two).
Every property runs over all models of the reference's own test data (tests/golden/cases) with a per-model seed."""
import math
import zlib

import numpy as np
import pytest

from tests import _cases as K
from yalps_amd import model as M
from yalps_amd import solve as S

SMALL = [n for n in K.names() if n not in K.LARGE]


def normal(model):
    """A case's model with constraints / variables / coefficients as lists of pairs (tests/helpers/read.ts)."""
    return {**model, "constraints": M.entries(model.get("constraints", {})),
            "variables": [(k, M.entries(v)) for k, v in M.entries(model.get("variables", {}))],
            "integers": list(model.get("integers") or []), "binaries": list(model.get("binaries") or [])}


def rng_of(name):
    return np.random.default_rng(zlib.crc32(name.encode()))


def same(a, b, variables=True):
    ta, tb = a.tableau, b.tableau
    assert (ta.width, ta.height, a.sign, a.integers) == (tb.width, tb.height, b.sign, b.integers)
    assert np.array_equal(ta.matrix + 0.0, tb.matrix + 0.0)  # (+0.0: the reference's deepEqual tests fold -0 into 0 first)
    assert np.array_equal(ta.position_of_variable, tb.position_of_variable)
    if variables:
        assert [k for k, _ in a.variables] == [k for k, _ in b.variables]


def num_rows(con):
    lo = con.get("equal", con.get("min"))
    hi = con.get("equal", con.get("max"))
    return (lo is not None) + (hi is not None)


def row_of_constraint(constraints, index):
    """First tableau row of the index-th constraint (distinct keys assumed, as in the reference's helper)."""
    return 1 + sum(num_rows(c) for _, c in constraints[:index])


@pytest.fixture(params=SMALL)
def case(request):
    c = K.load(request.param)
    return request.param, normal(c["model"])


def test_objective_can_share_a_key_with_a_constraint(case):  # tableau.ts:75-102
    name, model = case
    rng = rng_of(name)
    if not model["constraints"]:
        return
    idx = int(rng.integers(len(model["constraints"])))
    key, con = model["constraints"][idx]
    sign = 1.0 if ("equal" in con or "max" in con) else (-1.0 if "min" in con else 0.0)
    if sign == 0.0 or len({k for k, _ in model["constraints"]}) != len(model["constraints"]):
        return
    result = M.tableau_model({**model, "objective": key})
    expected = M.tableau_model(model)
    w, row = expected.tableau.width, row_of_constraint(model["constraints"], idx)
    expected.tableau.matrix[1:w] = expected.sign * sign * expected.tableau.matrix[row * w + 1:(row + 1) * w]
    same(result, expected)


def test_integers_and_binaries_as_bool_set_and_list(case):  # tableau.ts:135-183
    name, model = case
    keys = [k for k, _ in model["variables"]]
    rng = rng_of(name)
    sample = [k for k in keys if rng.random() < 0.5]
    for field in ("integers", "binaries"):
        none = [M.tableau_model({**model, field: v}) for v in (False, set(), [])]
        same(none[0], none[1]), same(none[2], none[1])
        every = [M.tableau_model({**model, field: v}) for v in (True, set(keys), keys)]
        same(every[0], every[1]), same(every[2], every[1])
        same(M.tableau_model({**model, field: sample}), M.tableau_model({**model, field: set(sample)}))


def test_binary_has_precedence_over_integer(case):  # tableau.ts:185-191
    name, model = case
    if not model["variables"]:
        return
    key = model["variables"][int(rng_of(name).integers(len(model["variables"])))][0]
    same(M.tableau_model({**model, "integers": [key], "binaries": [key]}),
         M.tableau_model({**model, "integers": [], "binaries": [key]}))


def test_swapping_a_bound_negates_its_row(case):  # tableau.ts:193-221
    name, model = case
    cons = model["constraints"]
    if len({k for k, _ in cons}) != len(cons):
        return
    one_sided = [i for i, (_, c) in enumerate(cons) if "equal" not in c and (("max" in c) != ("min" in c))]
    if not one_sided:
        return
    idx = one_sided[int(rng_of(name).integers(len(one_sided)))]
    key, con = cons[idx]
    swapped = {"min": con["max"]} if "min" not in con else {"max": con["min"]}
    new = list(cons)
    new[idx] = (key, swapped)
    result = M.tableau_model({**model, "constraints": new})
    expected = M.tableau_model(model)
    w, row = expected.tableau.width, row_of_constraint(cons, idx)
    expected.tableau.matrix[row * w:(row + 1) * w] *= -1.0
    same(result, expected)


def test_constraints_with_the_same_key_are_merged(case):  # tableau.ts:244-265
    name, model = case
    cons = model["constraints"]
    if not cons or len({k for k, _ in cons}) != len(cons):
        return
    rng = rng_of(name)
    idx = int(rng.integers(len(cons)))
    key, con = cons[idx]
    other = {"max": rng.random() * 100.0 + con.get("max", 0.0), "min": rng.random() * 100.0 + con.get("min", 0.0)}
    result = M.tableau_model({**model, "constraints": cons + [(key, other)]})
    merged = {"max": min(con.get("equal", con.get("max", math.inf)), other["max"]),
              "min": max(con.get("equal", con.get("min", -math.inf)), other["min"])}
    new = list(cons)
    new[idx] = (key, merged)
    same(result, M.tableau_model({**model, "constraints": new}))


def test_duplicate_variable_keys_and_last_coefficient_wins(case):  # tableau.ts:267-306
    name, model = case
    vars_ = model["variables"]
    if not vars_:
        return
    rng = rng_of(name)
    i_copy, i_change = int(rng.integers(len(vars_))), int(rng.integers(len(vars_)))
    renamed = list(vars_)
    renamed[i_change] = (vars_[i_copy][0], vars_[i_change][1])
    # (integers / binaries are looked up by key: the property holds for the matrix, as the reference states it)
    a, b = M.tableau_model({**model, "variables": renamed, "integers": [], "binaries": []}), \
        M.tableau_model({**model, "integers": [], "binaries": []})
    same(a, b, variables=False)
    vi = int(rng.integers(len(vars_)))
    vkey, coefs = vars_[vi]
    if coefs:
        ci = int(rng.integers(len(coefs)))
        ckey, value = coefs[ci]
        new_coefs = list(coefs)
        new_coefs[ci] = (ckey, value + rng.random() * 100.0)
        new_coefs.append((ckey, value))
        new_vars = list(vars_)
        new_vars[vi] = (vkey, new_coefs)
        if [k for k, _ in coefs].count(ckey) == 1:
            same(M.tableau_model({**model, "variables": new_vars}), M.tableau_model(model))


def test_removing_a_constraint_or_a_variable(case):  # tableau.ts:308-360
    name, model = case
    cons, vars_ = model["constraints"], model["variables"]
    rng = rng_of(name)
    base = M.tableau_model(model)
    t, w = base.tableau, base.tableau.width
    if cons and len({k for k, _ in cons}) == len(cons):
        idx = int(rng.integers(len(cons)))
        result = M.tableau_model({**model, "constraints": cons[:idx] + cons[idx + 1:]})
        row, gone = row_of_constraint(cons, idx), num_rows(cons[idx][1])
        keep = np.r_[0:row * w, (row + gone) * w:t.matrix.size]
        assert result.tableau.height == t.height - gone and np.array_equal(result.tableau.matrix, t.matrix[keep])
    if vars_ and len({k for k, _ in vars_}) == len(vars_):
        idx = int(rng.integers(len(vars_)))
        key = vars_[idx][0]
        result = M.tableau_model({**model, "variables": vars_[:idx] + vars_[idx + 1:]})
        full = t.matrix.reshape(t.height, w)
        rows = np.ones(t.height, bool)
        if key in set(model["binaries"]):  # its "x <= 1" row goes too
            binary_rows = [k for k, _ in vars_ if k in set(model["binaries"])]
            rows[t.height - len(binary_rows) + binary_rows.index(key)] = False
        expected = np.delete(full[rows], idx + 1, axis=1)
        assert result.tableau.width == w - 1 and np.array_equal(result.tableau.matrix.reshape(expected.shape), expected)


# ---- solve() (tests/solver.ts:49-124), with the CPU oracle as the simplex backend --------------------
@pytest.fixture(params=SMALL)
def solved(request, oracle):
    from tests.test_host_model import oracle_backend
    c = K.load(request.param)
    backend = oracle_backend(oracle)
    return request.param, c, backend, S._solve_with(backend, c["model"], c["options"])


def test_removing_unused_variables_keeps_the_optimum(solved):  # solver.ts:49-66
    name, c, backend, sol = solved
    model = normal(c["model"])
    if sol["status"] != "optimal" or len(model["variables"]) == len(sol["variables"]):
        return
    used = {k for k, _ in sol["variables"]}
    removed = S._solve_with(backend, {**model, "variables": [v for v in model["variables"] if v[0] in used]}, c["options"])
    assert K.valid_solution_and_status(removed, c["expected"], c["model"], c["options"])


def test_duplicating_a_non_binary_variable_keeps_the_optimum(solved):  # solver.ts:68-77
    name, c, backend, _ = solved
    model = normal(c["model"])
    non_binary = [v for v in model["variables"] if v[0] not in set(model["binaries"])]
    if not non_binary:
        return
    pick = non_binary[int(rng_of(name).integers(len(non_binary)))]
    dup = S._solve_with(backend, {**model, "variables": model["variables"] + [pick]}, c["options"])
    # (the duplicate shares its key: only the objective and the status are compared, as resultIsOptimal does)
    assert dup["status"] == c["expected"]["status"]
    assert K.result_is_optimal(dup["result"], c["expected"]["result"], c["options"])


def test_tolerance_option_gives_a_result_in_range(solved):  # solver.ts:114-124
    name, c, backend, _ = solved
    model = normal(c["model"])
    if not model["integers"] and not model["binaries"]:
        return
    tol = c["options"]["tolerance"]
    options = {**c["options"], "tolerance": float(rng_of(name).random()) * (1.0 - tol) + tol}
    sol = S._solve_with(backend, c["model"], options)
    assert K.valid_solution_and_status(sol, c["expected"], c["model"], options)
