"""The reference's benchmark harness (/root/reference/benchmarks/benchmark.ts:64-126) restated:
runners with convert / solve / value, `num_samples` timed solves per runner without outlier
rejection, Kahan-Babuska-Neumaier mean and sample variance, and the mean / stdDev / slowdown table
sorted by mean.  Host-side measurement code; the solvers it times are the MI355X paths.
"""
import gc
import math
import time
from dataclasses import dataclass
from typing import Any, Callable

from .solve import solve

MAX_DIFF = 1e-5  # tests/helpers/validate.ts:4


@dataclass(frozen=True)
class Runner:
    """benchmark.ts:6-11"""
    name: str
    convert: Callable[[Any, dict], Any]
    solve: Callable[[Any], Any]
    value: Callable[[Any], float]


def kahan_babushka_neumaier_sum(values):
    """benchmark.ts:31-40"""
    total, c = 0.0, 0.0
    for value in values:
        t = total + value
        c += (total - t + value) if abs(total) >= abs(value) else (value - t + total)
        total = t
    return total + c


def stats(samples):
    """benchmark.ts:49-53: mean and (n-1)-normalised variance"""
    mean = kahan_babushka_neumaier_sum(samples) / len(samples)
    variance = kahan_babushka_neumaier_sum([(x - mean) * (x - mean) for x in samples]) / (len(samples) - 1)
    return {"mean": mean, "variance": variance}


def _time_ms(runner, inp):
    """benchmark.ts:55-60 (performance.now() is in milliseconds)"""
    start = time.perf_counter()
    runner.solve(inp)
    return (time.perf_counter() - start) * 1e3


def sample_benchmark(solvers, bench, num_samples):
    """benchmark.ts:64-80"""
    data = []
    for runner in solvers:
        inp = runner.convert(bench["model"], bench["options"])
        gc.collect()  # isolate time due to gc between solvers (:70)
        times = [_time_ms(runner, inp) for _ in range(num_samples)]  # outliers are kept (:72)
        data.append((runner.name, stats(times)))
    return data


def format_num(x):
    """benchmark.ts:82 parseFloat(x.toFixed(2))"""
    return float("%.2f" % x) if math.isfinite(x) else x


def results_table(results):
    """benchmark.ts:84-97: rows sorted by mean; slowdown relative to the fastest"""
    rows = sorted(results, key=lambda r: r[1]["mean"])
    fastest = rows[0][1]["mean"]
    return {name: {"mean": format_num(s["mean"]), "stdDev": format_num(math.sqrt(s["variance"])),
                   "slowdown": format_num(s["mean"] / fastest)} for name, s in rows}


def result_is_optimal(result, expected, options):
    """tests/helpers/validate.ts:6-16"""
    if math.isnan(expected):
        return math.isnan(result)
    if math.isinf(expected):
        return expected == result
    rel = (abs(result - expected) - options["precision"]) / max(abs(expected), 1.0)
    return math.isfinite(result) and rel <= max(options["tolerance"], MAX_DIFF)


def validate(bench, runner):
    """benchmark.ts:99-104"""
    result = runner.value(runner.solve(runner.convert(bench["model"], bench["options"])))
    assert result_is_optimal(result, bench["expected"], bench["options"]), (bench["name"], runner.name, result)


def _count(model, key):
    v = model.get(key)
    return 0 if v is None or isinstance(v, bool) else len(list(v))


def benchmark(benchmarks, solvers, num_samples=30, run_validation=True, out=print):
    """benchmark.ts:106-126.  Also returns [(headline, table)] for machine use."""
    from .model import entries
    tables = []
    for bench in benchmarks:
        if run_validation:
            for runner in solvers:
                validate(bench, runner)
        model = bench["model"]
        head = "%s: %d constraints, %d variables, %d integers:" % (
            bench["name"], len({k for k, _ in entries(model.get("constraints", {}))}),
            len(entries(model.get("variables", {}))), _count(model, "integers") + _count(model, "binaries"))
        table = results_table(sample_benchmark(solvers, bench, num_samples))
        out(head)
        w = max(len(n) for n in table) + 2
        out("%-*s %10s %10s %10s" % (w, "(index)", "mean", "stdDev", "slowdown"))
        for name, row in table.items():
            out("%-*s %10s %10s %10s" % (w, name, row["mean"], row["stdDev"], row["slowdown"]))
        out("")
        tables.append((head, table))
    return tables


def _with_infinite_pivots(options):
    return {**options, "maxPivots": math.inf}  # benchmarks/runners.ts:10


# benchmarks/runners.ts:8-13, one runner per boundary of this build
hip_runner = Runner("YALPS-hip", lambda model, options: (model, _with_infinite_pivots(options)),
                    lambda inp: solve(inp[0], inp[1]), lambda s: s["result"])
hip_dense_runner = Runner("YALPS-hip (host tableau, host nodes)", lambda model, options: (model, _with_infinite_pivots(options)),
                          lambda inp: solve(inp[0], inp[1], sparse=False, device_nodes=False, native=False), lambda s: s["result"])
hip_batched_runner = Runner("YALPS-hip (one node at a time)", lambda model, options: (model, _with_infinite_pivots(options)),
                            lambda inp: solve(inp[0], inp[1], node_batch=0), lambda s: s["result"])
runners = (hip_runner, hip_dense_runner, hip_batched_runner)
