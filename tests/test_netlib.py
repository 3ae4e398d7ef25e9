"""BASELINE config 3: the netlib problems the reference benchmarks itself on (benchmarks/netlib/
read.ts filters -> 25 problems, tests/additional/netlib.ts), read by yalps_amd.mps from the
reference's own data files (tests/golden/netlib) and checked with the reference's tolerance
(tests/helpers/validate.ts: relative 1e-5)."""
import math

import pytest

from tests import _cases as K
from tests import _golden as G
from yalps_amd import model as M
from yalps_amd import mps
from yalps_amd import solve as S

DIR = __import__("os").path.join(G.GOLDEN, "netlib")
# tableau h x w per SURVEY.md appendix B (as built by the reference's tableauModel)
SHAPES = {"AGG2": (577, 303), "AGG3": (577, 303), "BEACONFD": (314, 263), "ISRAEL": (175, 143), "LOTFI": (249, 309),
          "SC105": (151, 104), "SC205": (297, 204), "SCAGR25": (772, 501), "SCAGR7": (214, 141), "SCFXM1": (518, 458),
          "SCORPION": (669, 359), "SCRS8": (875, 1170), "SCSD6": (295, 1351), "SCTAP1": (421, 481),
          "SCTAP2": (1561, 1881), "SCTAP3": (2101, 2481), "SHARE1B": (207, 226), "SHIP04L": (757, 2119),
          "SHIP04S": (757, 1459), "SHIP08L": (1477, 4284), "SHIP08S": (1477, 2388), "SHIP12L": (2197, 5428),
          "SHIP12S": (2197, 2764), "STOCFOR1": (181, 112), "KLEIN2": (478, 55)}
SMALL = ("SC105", "SC205", "SCAGR7", "STOCFOR1", "LOTFI", "SHARE1B", "ISRAEL", "KLEIN2", "BEACONFD")


@pytest.fixture(scope="module")
def benchmarks():
    return {b["name"]: b for b in mps.read_benchmarks(DIR)}


def _ok(sol, b):
    if math.isnan(b["expected"]):  # KLEIN2: index value null -> not optimal (infeasible)
        return math.isnan(sol["result"])
    return sol["status"] == "optimal" and K.result_is_optimal(sol["result"], b["expected"], b["options"])


def test_reader_selects_the_reference_subset(benchmarks):
    assert set(benchmarks) == set(SHAPES)  # netlib/read.ts `ok` list
    for name, b in benchmarks.items():
        t = M.tableau_model(b["model"]).tableau
        assert (t.height, t.width) == SHAPES[name], name
    assert benchmarks["KLEIN2"]["options"]["checkCycles"] is True


def _ln(f1="", f2="", f3="", f4="", f5="", f6=""):
    """One fixed-column MPS data line (fields at columns 2-3, 5-12, 15-22, 25-36, 40-47, 50-61)."""
    return (" %-2s %-8s  %-8s  %12s   %-8s  %12s" % (f1, f2, f3, f4, f5, f6)).rstrip()


def test_mps_reader_details():
    text = "\n".join([
        "NAME          TINY", "ROWS", _ln("N", "COST"), _ln("L", "LIM1"), _ln("G", "MYEQN"), _ln("E", "FIX"), "COLUMNS",
        _ln("", "X", "COST", "1.0", "LIM1", "1.0"), _ln("", "X", "MYEQN", "1.0"),
        _ln("", "Y", "COST", "2.0", "FIX", "-1.0"), "* a comment", "RHS",
        _ln("", "RHS", "LIM1", "4.0", "MYEQN", "1.0"), _ln("", "RHS", "FIX", "7.0"), "RANGES",
        _ln("", "RNG", "LIM1", "2.5", "FIX", "-3.0"), "ENDATA", ""])
    m = mps.model_from_mps(text, "minimize")
    assert (m["name"], m["objective"]) == ("TINY", "COST")
    assert m["constraints"]["LIM1"] == [1.5, 4.0] and m["constraints"]["MYEQN"] == [1.0, math.inf]
    assert m["constraints"]["FIX"] == [4.0, 7.0]  # E row with a negative range: [rhs - |R|, rhs]
    assert m["variables"] == {"X": {"COST": 1.0, "LIM1": 1.0, "MYEQN": 1.0}, "Y": {"COST": 2.0, "FIX": -1.0}}
    assert mps.convert_constraints(m["constraints"]) == {"LIM1": {"min": 1.5, "max": 4.0}, "MYEQN": {"min": 1.0},
                                                         "FIX": {"min": 4.0, "max": 7.0}}
    with pytest.raises(mps.MPSError):
        mps.model_from_mps("ROWS\n", None)


@pytest.mark.parametrize("name", SMALL)
def test_netlib_small_with_oracle_backend(oracle, benchmarks, name):
    from tests.test_host_model import oracle_backend
    b = benchmarks[name]
    sol = S._solve_with(oracle_backend(oracle), b["model"], b["options"])
    assert _ok(sol, b), (sol["status"], sol["result"], b["expected"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_netlib_on_gpu(benchmarks, name):
    """All 25 problems through solve() with the HIP simplex; objective within the reference's
    tolerance of the published netlib optimum (index.json)."""
    b = benchmarks[name]
    sol = S.solve(b["model"], b["options"])
    assert _ok(sol, b), (sol["status"], sol["result"], b["expected"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("SC205", "SCAGR25", "SHARE1B", "KLEIN2", "SCTAP2"))
def test_netlib_gpu_equals_oracle(oracle, benchmarks, name):
    from tests.test_host_model import oracle_backend
    b = benchmarks[name]
    sol = S.solve(b["model"], b["options"])
    ref = S._solve_with(oracle_backend(oracle), b["model"], b["options"])
    assert sol["status"] == ref["status"] and G.same_number(sol["result"], ref["result"])
    assert sol["variables"] == ref["variables"]
