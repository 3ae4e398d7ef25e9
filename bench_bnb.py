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

Under torch.distributed.run (--gpus N, one process per GPU) the node queue is dealt round-robin to the
ranks -- nodes are independent given the root, so there is NO data-path collective; the
group only carries the barriers and the max-over-ranks time, and `value` is the whole-job rate.
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
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
        # (ranks sharing one GPU, as in the tests on a 1-GPU box, cannot use RCCL: gloo carries the barriers)
        dist.init_process_group("nccl" if ndev >= world else "gloo",
                                **({"device_id": torch.device("cuda", device)} if ndev >= world else {}))
    else:
        device = 0

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

    ctx = _native.Context(device)
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
    all_nodes = len(nodes)
    nodes = nodes[rank::world]  # this rank's share of the queue (every rank built the same queue)

    def barrier():
        if dist is not None:
            dist.barrier()

    # (a) one batch
    batch.solve(nodes, 1e-8, 8192)  # warm-up
    barrier()
    t0 = time.perf_counter()
    sts, ress, pivs, heights, gpu_ms = batch.solve(nodes, 1e-8, 8192)
    wall_batch = time.perf_counter() - t0
    barrier()
    if dist is not None:
        import torch
        tm = torch.tensor([wall_batch, gpu_ms], dtype=torch.float64)
        tm = tm.cuda() if dist.get_backend() == "nccl" else tm
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall_batch, gpu_ms = float(tm[0]), float(tm[1])

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

    # oracle check on a sample (bit-exact whole tableau); the oracle's own time on those nodes (applyCuts on the host +
    # simplex, one core) is the CPU baseline of this config
    orc = _oracle.load()
    step = max(1, len(nodes) // max(args.check, 1))
    checked = 0
    cpu_s, cpu_pivots = 0.0, 0
    for i in range(0, len(nodes), step):
        t0 = time.perf_counter()
        cur = BC.apply_cuts(root, buf, nodes[i])
        mm, pp, vv = cur.matrix.copy(), cur.position_of_variable.copy(), cur.variable_at_position.copy()
        t1 = time.perf_counter()
        est, eres, epiv, _ = orc.simplex(mm, cur.width, cur.height, pp, vv)
        cpu_s += time.perf_counter() - t1
        cpu_pivots += epiv
        gm, _, gp, gv = batch.download(i, cur.height, matrix=True)
        assert est == sts[i] and epiv == int(pivs[i]) and np.array_equal(gm.view(np.int64), mm.view(np.int64))
        assert np.array_equal(gp, pp) and np.array_equal(gv, vv)
        checked += 1
    batch.close()
    ctx.close()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    total_piv = int(pivs.sum())
    bytes_alg = sum(16 * int(heights[i]) * w * int(pivs[i]) for i in range(len(nodes)))
    print(json.dumps({
        "metric": "branch-and-cut nodes/sec (batched node LPs, %d MI355X)" % world, "value": all_nodes / (gpu_ms * 1e-3),
        "unit": "nodes/s", "n_gpus": world, "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d nodes of a %d-variable x %d-row dense MILP (root tableau %dx%d), <= %d cuts per node"
                               % (all_nodes, N, Mr, h, w, max_cuts),
                   "nodes_per_rank": len(nodes), "collectives_on_the_data_path": 0},
        "batch": {"gpu_ms": gpu_ms, "wall_ms": 1e3 * wall_batch, "pivots_rank0": total_piv, "pivots_per_s_rank0": total_piv / (gpu_ms * 1e-3),
                  "algorithmic_GBps_rank0": bytes_alg / (gpu_ms * 1e-3) / 1e9,
                  "status_counts": {s: sts.count(s) for s in sorted(set(sts))}},
        "one_at_a_time": {"nodes": k, "wall_ms": 1e3 * wall_seq, "nodes_per_s": k / wall_seq},
        "speedup_vs_one_at_a_time": (all_nodes / wall_batch) / (k / wall_seq),
        "oracle_checked_nodes": checked,
        "cpu_baseline": {"value": checked / cpu_s, "unit": "nodes/s", "cores": 1, "kind": "port",
                         "pivots_per_s": cpu_pivots / cpu_s,
                         "sample": "%d of this rank's nodes (every %d-th), oracle/simplex_oracle.c simplex() on the host-built node "
                                   "tableau (applyCuts not timed), %.2f s, host has %d cores" % (checked, step, cpu_s, os.cpu_count())}}))


if __name__ == "__main__":
    main()
