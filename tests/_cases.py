"""Loader + validator for tests/golden/cases/*.json (data files of the reference's own test-suite).
The validator restates /root/reference/tests/helpers/validate.ts (the reference's parity
definition: status equality, objective within rel 1e-5, feasibility, integrality) and the reader
/root/reference/tests/helpers/read.ts:41-60."""
import json
import math
import os

from yalps_amd.model import entries
from yalps_amd.solve import default_options

CASES_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases")
LARGE = ("Monster 2", "Monster Problem", "Vendor Selection")  # read.ts:39
MAX_DIFF = 1e-5  # validate.ts:4


def names():
    return sorted(f[:-5] for f in os.listdir(CASES_DIR) if f.endswith(".json"))


def load(name):
    with open(os.path.join(CASES_DIR, name + ".json")) as f:
        data = json.load(f)
    model = data["model"]
    options = dict(default_options)
    options.update(data.get("options") or {})
    exp = dict(data["expected"])
    if exp["status"] == "optimal":  # read.ts:54-57
        result = float(exp["result"])
    elif exp["status"] == "unbounded":
        result = math.inf * (-1.0 if model.get("direction") == "minimize" else 1.0)
    else:
        result = math.nan
    exp["result"] = result
    return {"name": name, "model": model, "options": options, "expected": exp}


def _rel_from(delta, expected, precision):  # validate.ts:6-7
    return (delta - precision) / max(abs(expected), 1.0)


def _rel(result, expected, precision):  # :9-10
    return _rel_from(abs(result - expected), expected, precision)


def result_is_optimal(result, expected, options):  # :13-16
    if math.isnan(expected):
        return math.isnan(result)
    if math.isinf(expected):
        return expected == result
    return math.isfinite(result) and _rel(result, expected, options["precision"]) <= max(options["tolerance"], MAX_DIFF)


def constraints_are_satisfied(solution, model, precision):  # :18-41
    variables = dict(entries(model["variables"]))
    sums = {}
    for key, num in solution["variables"]:
        for constraint, coef in entries(variables[key]):
            sums[constraint] = num * coef + sums.get(constraint, 0.0)
    for key, con in entries(model["constraints"]):
        s = sums.get(key, 0.0)
        eq, mn, mx = con.get("equal"), con.get("min"), con.get("max")
        if eq is not None:
            if _rel(s, eq, precision) > MAX_DIFF:
                return False
        else:
            if mn is not None and _rel_from(mn - s, mn, precision) > MAX_DIFF:
                return False
            if mx is not None and _rel_from(s - mx, mx, precision) > MAX_DIFF:
                return False
    return True


def variables_have_valid_values(solution, model, precision):  # :43-53
    ints, bins = set(model.get("integers") or ()), set(model.get("binaries") or ())
    for key, n in solution["variables"]:
        if not n >= -precision:
            return False
        if (key in ints or key in bins) and not abs(n - math.floor(n + 0.5)) <= precision:
            return False
        if key in bins and not n <= 1 + precision:
            return False
    return True


def valid_solution(solution, expected_result, model, options):  # :55-63
    return (result_is_optimal(solution["result"], expected_result, options)
            and variables_have_valid_values(solution, model, options["precision"])
            and (not math.isfinite(expected_result) or constraints_are_satisfied(solution, model, options["precision"])))


def valid_solution_and_status(solution, expected, model, options):  # :65-74
    if solution["status"] != expected["status"]:
        return False
    if solution["status"] == "timedout" and math.isnan(solution["result"]):
        return True
    return valid_solution(solution, expected["result"], model, options)
