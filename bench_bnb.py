#!/usr/bin/env python3
"""bench_bnb.py -- BASELINE config 4: a ~1024-node queue of branch-and-cut sub-problems of a
512-variable MILP evaluated on 1 MI355X.  (The headline metric lives in bench.py; this is the
measurement of the batched node path)

A synthetic dense MILP (maximize c.x, A x <= b, x integer; the same PRNG as the dense-LP generator)
is solved at the root on the GPU; a breadth-first expansion on the most fractional variable
(src/branchAndCut.ts:64-85,141-156) produces the node queue; then the SAME 1024 nodes are timed
  (a) as one batch (yalps_batch_solve: one workgroup per node, root resident, cuts applied on device),
  (b) one at a time through the drop-in call (applyCuts on the host, upload, solve, download), on a sample,
and a sample of nodes is checked bit for bit against the CPU oracle.  Prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vars", type=int, default=512)
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--seq-sample", type=int, default=128)
    ap.add_argument("--check", type=int, default=32)
    args = ap.parse_args()

    from tests import _oracle
    from yalps_amd import _native, branch_and_cut as BC
    from yalps_amd.model import Tableau
    N, Mr = args.vars, args.rows
    w, h = N + 1, Mr + 1
    m = _native.dense_lp(Mr, N, 4242)
    ident = np.arange(w + h, dtype=np.int32)
    pos, var = ident.copy(), ident.copy()
    st, res, _ = _native.simplex_host(m, w, h, pos, var, max_pivots=math.inf)
    assert st == "optimal"
    root = Tableau(m, w, h, pos, var)
    ints = list(range(1, N + 1))
    max_cuts = 24

    ctx = _native.Context(0)
    batch = _native.NodeBatch(ctx, w, h, max_cuts, args.nodes)
    batch.set_root(m, pos, var)

    # breadth-first node queue
    variable, value, frac = BC.most_fractional_var(root, ints)
    frontier = [((-1, variable, float(math.ceil(value))),), ((1, variable, float(math.floor(value))),)]
    nodes = []
    while len(nodes) < args.nodes and frontier:
        take = frontier[: args.nodes - len(nodes)]
        frontier = frontier[len(take):]
        sts, ress, pivs, heights, _ = batch.solve(take, 1e-8, 8192)
        for i, cuts in enumerate(take):
            nodes.append(cuts)
            if sts[i] == "optimal" and len(cuts) < max_cuts:
                _, col0, p, v = batch.download(i, int(heights[i]))
                view = Tableau(None, w, int(heights[i]), p, v, col0)
                variable, value, frac = BC.most_fractional_var(view, ints)
                if frac > 1e-8:
                    upper = tuple(c for c in cuts if not (c[1] == variable and c[0] < 0)) + ((-1, variable, float(math.ceil(value))),)
                    lower = tuple(c for c in cuts if not (c[1] == variable and c[0] > 0)) + ((1, variable, float(math.floor(value))),)
                    frontier += [upper, lower]
    nodes = nodes[: args.nodes]

    # (a) one batch
    batch.solve(nodes, 1e-8, 8192)  # warm-up
    t0 = time.perf_counter()
    sts, ress, pivs, heights, gpu_ms = batch.solve(nodes, 1e-8, 8192)
    wall_batch = time.perf_counter() - t0

    # (b) one node at a time through the drop-in (host applyCuts + upload + solve + download)
    k = min(args.seq_sample, len(nodes))
    buf = (np.zeros(m.size + max_cuts * w), np.zeros(w + h + max_cuts, np.int32), np.zeros(w + h + max_cuts, np.int32))
    t0 = time.perf_counter()
    seq = []
    for cuts in nodes[:k]:
        cur = BC.apply_cuts(root, buf, cuts)
        s_, r_, p_ = _native.simplex_host(cur.matrix, cur.width, cur.height, cur.position_of_variable,
                                          cur.variable_at_position, max_pivots=8192)
        seq.append((s_, r_, p_))
    wall_seq = time.perf_counter() - t0
    for i in range(k):
        assert seq[i][0] == sts[i] and seq[i][2] == int(pivs[i])
        assert (math.isnan(seq[i][1]) and math.isnan(ress[i])) or seq[i][1] == ress[i]

    # oracle check on a sample (bit-exact whole tableau)
    orc = _oracle.load()
    step = max(1, len(nodes) // max(args.check, 1))
    checked = 0
    for i in range(0, len(nodes), step):
        cur = BC.apply_cuts(root, buf, nodes[i])
        mm, pp, vv = cur.matrix.copy(), cur.position_of_variable.copy(), cur.variable_at_position.copy()
        est, eres, epiv, _ = orc.simplex(mm, cur.width, cur.height, pp, vv)
        gm, _, gp, gv = batch.download(i, cur.height, matrix=True)
        assert est == sts[i] and epiv == int(pivs[i]) and np.array_equal(gm.view(np.int64), mm.view(np.int64))
        assert np.array_equal(gp, pp) and np.array_equal(gv, vv)
        checked += 1
    batch.close()
    ctx.close()

    total_piv = int(pivs.sum())
    bytes_alg = sum(16 * int(heights[i]) * w * int(pivs[i]) for i in range(len(nodes)))
    print(json.dumps({
        "metric": "branch-and-cut nodes/sec (batched node LPs, 1 MI355X)", "value": len(nodes) / (gpu_ms * 1e-3),
        "unit": "nodes/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d nodes of a %d-variable x %d-row dense MILP (root tableau %dx%d), <= %d cuts per node"
                               % (len(nodes), N, Mr, h, w, max_cuts)},
        "batch": {"gpu_ms": gpu_ms, "wall_ms": 1e3 * wall_batch, "pivots": total_piv, "pivots_per_s": total_piv / (gpu_ms * 1e-3),
                  "algorithmic_GBps": bytes_alg / (gpu_ms * 1e-3) / 1e9,
                  "status_counts": {s: sts.count(s) for s in sorted(set(sts))}},
        "one_at_a_time": {"nodes": k, "wall_ms": 1e3 * wall_seq, "nodes_per_s": k / wall_seq},
        "speedup_vs_one_at_a_time": (len(nodes) / wall_batch) / (k / wall_seq),
        "oracle_checked_nodes": checked}))


if __name__ == "__main__":
    main()
