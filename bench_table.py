#!/usr/bin/env python3
"""bench_table.py -- the reference README's benchmark table (README.md:276-373, whole-solve() times of
11 problems) re-measured with the MI355X core behind the same API (lite).

For every problem: the model is read from the reference's own data files (tests/golden), the
tableau is built by the Python mirror of tableauModel (NOT timed: in an integration that stays the
TypeScript host's job), and what is timed is what this build replaces: the simplex() calls --
  lp_ms     one drop-in call yalps_simplex_f64 on host arrays (upload + solve + download), root LP
  lp_sparse_ms  the same LP through yalps_simplex_sparse_f64 (cells up, column 0 + permutations back)
  milp_ms   the whole branch and cut: sequential (one drop-in call per node), batched (node_batch=32), and with root
            and nodes resident in HBM (device_nodes: yalps_tableau_apply_cuts); native = the whole branch and cut in one
            native call (yalps_milp_f64), one node at a time or in batches of 32
next to the reference's published whole-solve() mean (unknown CPU, node 19) for orientation only.
Then the reference's own harness (benchmarks/benchmark.ts, restated in yalps_amd/benchmark.py: 30
samples per runner, compensated mean / stdDev / slowdown, validated results) over the same problems
with whole-solve() runners -- these include the Python host (tableau_model, heap, cuts), which in an
integration is the TypeScript host's work.  Tables go to stderr; prints one JSON object.
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

README_MS = {"Monster 2": 53.95, "Monster Problem": 1.85, "Vendor Selection": 296.05, "Large Farm MIP": 30.46,
             "AGG2": 1.6, "BEACONFD": 2.59, "SC205": 7.18, "SCFXM1": 20.67, "SCRS8": 56.8, "SCTAP2": 49.98,
             "SHIP08S": 17.86}


def main():
    from tests import _cases as K, _golden as G, _oracle
    from tests.test_host_model import oracle_backend
    from yalps_amd import _native, model as M, mps, solve as S
    orc = _oracle.load()
    netlib = {b["name"]: b for b in mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib"))}
    rows, benches = [], []
    for name, ref_ms in README_MS.items():
        if name in netlib:
            b = netlib[name]
            mdl, opt, expected = b["model"], dict(b["options"]), b["expected"]
        else:
            c = K.load(name)
            mdl, opt, expected = c["model"], dict(c["options"]), c["expected"]["result"]
        benches.append({"name": name, "model": mdl, "options": {**S.default_options, **opt}, "expected": expected})
        opt["maxPivots"] = math.inf  # benchmarks/runners.ts:10
        tm = M.tableau_model(mdl)
        t = tm.tableau
        lp = []
        for _ in range(5):
            m, pos, var = t.matrix.copy(), t.position_of_variable.copy(), t.variable_at_position.copy()
            t0 = time.perf_counter()
            st, res, piv = _native.simplex_host(m, t.width, t.height, pos, var, precision=opt["precision"],
                                                max_pivots=opt["maxPivots"], check_cycles=opt["checkCycles"])
            lp.append(time.perf_counter() - t0)
        # the same root LP on one host core: the CPU oracle (the scalar port of src/simplex.ts) on the same initial tableau
        m, pos, var = t.matrix.copy(), t.position_of_variable.copy(), t.variable_at_position.copy()
        t0 = time.perf_counter()
        cst, cres, cpiv, _ = orc.simplex(m, t.width, t.height, pos, var, precision=opt["precision"], max_pivots=opt["maxPivots"],
                                         check_cycles=opt["checkCycles"])
        cpu_lp = time.perf_counter() - t0
        assert (cst, cpiv) == (st, piv)
        cells = M.tableau_model(mdl, sparse=True).tableau.cells
        sp = []
        for _ in range(5):
            t0 = time.perf_counter()
            st2, res2, piv2, _, _, _ = _native.simplex_sparse(t.width, t.height, *cells, precision=opt["precision"],
                                                              max_pivots=opt["maxPivots"], check_cycles=opt["checkCycles"])
            sp.append(time.perf_counter() - t0)
        assert (st2, piv2) == (st, piv) and (res2 == res or (res2 != res2 and res != res))
        row = {"problem": name, "tableau": "%dx%d" % (t.height, t.width), "integers": len(tm.integers),
               "root_status": st, "root_pivots": piv, "lp_ms": round(1e3 * min(lp), 3), "lp_sparse_ms": round(1e3 * min(sp), 3),
               "cells": int(cells[0].size),
               "us_per_pivot": round(1e6 * min(lp) / max(piv, 1), 2), "reference_solve_ms_readme": ref_ms,
               "cpu_baseline": {"value": round(1e3 * cpu_lp, 3), "unit": "ms per simplex() of the root LP", "cores": 1, "kind": "port",
                                "sample": "oracle/simplex_oracle.c on the same initial tableau, %d pivots" % cpiv}}
        if tm.integers:
            for label, nb, dev, nat in (("milp_sequential_ms", 0, False, False), ("milp_batched_ms", 32, False, False),
                                        ("milp_device_nodes_ms", 0, True, False), ("milp_native_ms", 0, True, True),
                                        ("milp_native_batched_ms", 32, True, True)):
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    sol = S.solve(mdl, opt, node_batch=nb, device_nodes=dev, native=nat)
                    ts.append(time.perf_counter() - t0)
                row[label] = round(1e3 * min(ts), 2)  # includes the Python host (tableau build, heap, cuts)
            # the whole branch and cut with every node LP on one host core: the same Python driver as milp_sequential_ms
            # around the CPU oracle; cpu_simplex_ms = the time inside the oracle alone (what the GPU replaces)
            inside = [0.0, 0]

            def cpu_backend(tableau, options, _b=oracle_backend(orc)):
                t0 = time.perf_counter()
                out = _b(tableau, options)
                inside[0] += time.perf_counter() - t0
                inside[1] += 1
                return out
            t0 = time.perf_counter()
            csol = S._solve_with(cpu_backend, mdl, opt)
            row["milp_cpu_sequential_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
            row["cpu_baseline_milp"] = {"value": round(1e3 * inside[0], 2), "unit": "ms inside simplex() over the whole branch and cut",
                                        "cores": 1, "kind": "port", "sample": "%d simplex() calls (root + nodes), oracle/simplex_oracle.c" % inside[1]}
            assert csol["status"] == sol["status"]
            row["objective_ok"] = bool(K.result_is_optimal(sol["result"], expected, S.default_options | {"tolerance": opt.get("tolerance", 0)}))
        rows.append(row)
    from yalps_amd import benchmark as B
    samples = int(os.environ.get("YALPS_BENCH_SAMPLES", "30"))
    tables = B.benchmark(benches, B.runners, num_samples=samples, out=lambda line: print(line, file=sys.stderr))
    print(json.dumps({"what": "simplex() time behind solve() on 1 MI355X vs the reference README table", "rows": rows,
                      "harness": {"samples": samples, "unit": "ms per solve() incl. the Python host",
                                  "tables": [{"benchmark": h, "results": t} for h, t in tables]}}))


if __name__ == "__main__":
    main()
