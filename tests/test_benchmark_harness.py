"""The benchmark harness restated from the reference (benchmarks/benchmark.ts:31-97): summation,
statistics and the table layout, on CPU with stub runners."""
import math

import numpy as np
import pytest

from yalps_amd import benchmark as B


def test_compensated_sum_beats_naive():
    vals = [1e16, 1.0, -1e16, 1.0] * 1000
    assert B.kahan_babushka_neumaier_sum(vals) == 2000.0
    assert B.kahan_babushka_neumaier_sum([]) == 0.0
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 100, 999).tolist()
    assert abs(B.kahan_babushka_neumaier_sum(x) - math.fsum(x)) <= 1e-10


def test_stats_is_sample_variance():
    rng = np.random.default_rng(5)
    x = rng.uniform(1, 9, 30)
    s = B.stats(x.tolist())
    assert abs(s["mean"] - x.mean()) < 1e-12 and abs(s["variance"] - x.var(ddof=1)) < 1e-12


def test_format_num_is_two_decimals():
    assert B.format_num(1.005) == 1.0 and B.format_num(2.346) == 2.35 and B.format_num(17.0) == 17.0


def test_results_table_sorted_with_slowdown():
    t = B.results_table([("slow", {"mean": 30.0, "variance": 4.0}), ("fast", {"mean": 10.0, "variance": 1.0})])
    assert list(t) == ["fast", "slow"]
    assert t["fast"] == {"mean": 10.0, "stdDev": 1.0, "slowdown": 1.0}
    assert t["slow"] == {"mean": 30.0, "stdDev": 2.0, "slowdown": 3.0}


def test_benchmark_runs_validation_and_sampling():
    calls = []
    stub = B.Runner("stub", lambda m, o: (m, o), lambda inp: calls.append(1) or {"result": 14400.0}, lambda s: s["result"])
    bench = {"name": "toy", "model": {"constraints": {"a": {"max": 1}}, "variables": {"x": {"a": 1}, "y": {"a": 2}}, "integers": ["x"]},
             "options": {"precision": 1e-8, "tolerance": 0}, "expected": 14400.0}
    lines = []
    tables = B.benchmark([bench], [stub], num_samples=7, out=lines.append)
    assert len(calls) == 8  # 1 validation + 7 samples
    assert lines[0] == "toy: 1 constraints, 2 variables, 1 integers:"
    assert tables[0][1]["stub"]["slowdown"] == 1.0
    bad = B.Runner("bad", lambda m, o: None, lambda inp: {"result": 1.0}, lambda s: s["result"])
    try:
        B.benchmark([bench], [bad], num_samples=2, out=lines.append)
        raise SystemExit("validation should have failed")
    except AssertionError:
        pass


@pytest.mark.gpu
def test_harness_with_the_hip_runners_on_readme_problems():
    """benchmarks/benchmark.ts:98-126 with the MI355X runners on four of the reference README's problems (two MILPs, one
    LP from the test-suite's data, one netlib LP): every runner's result is validated against the expected optimum first
    (validate(), relative 1e-5 / the model's tolerance), then sampled; the table has one row per runner, sorted by mean,
    slowdown 1 for the fastest."""
    import os

    from tests import _cases as K, _golden as G
    from yalps_amd import mps, solve as S
    netlib = {b["name"]: b for b in mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib"))}
    benches = []
    for name in ("Monster Problem", "Large Farm MIP", "Knapsack 1", "SC205"):
        if name in netlib:
            b = netlib[name]
            mdl, opt, expected = b["model"], dict(b["options"]), b["expected"]
        else:
            c = K.load(name)
            mdl, opt, expected = c["model"], dict(c["options"]), c["expected"]["result"]
        benches.append({"name": name, "model": mdl, "options": {**S.default_options, **opt}, "expected": expected})
    lines = []
    tables = B.benchmark(benches, B.runners, num_samples=5, out=lines.append)
    assert len(tables) == 4 and all(len(t) == len(B.runners) for _, t in tables)
    for head, table in tables:
        means = [row["mean"] for row in table.values()]
        assert means == sorted(means) and list(table.values())[0]["slowdown"] == 1.0 and all(m > 0 for m in means), (head, table)
    assert lines[0].startswith("Monster Problem: 600 constraints, 552 variables, 0 integers")
    # a runner that returns a wrong optimum must be caught by the validation pass, before anything is timed
    wrong = B.Runner("wrong", B.hip_runner.convert, B.hip_runner.solve, lambda s: s["result"] * 1.01 + 1.0)
    with pytest.raises(AssertionError):
        B.benchmark(benches[:1], [wrong], num_samples=2, out=lines.append)
