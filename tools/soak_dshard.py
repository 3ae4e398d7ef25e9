"""Randomised soak of the row shards with delayed row updates (dshard_kernel / dshard_select_kernel) against the
CPU oracle (test infrastructure, like tests/): one-rank shards of seeded random tableaux -- random shape, sparsity, signs of
the right-hand sides (phase-1 starts), degenerate rows, pivot budget, delay depth, cache policy; the Python driver with the
status read back every 1..8 pivots.  Every solve must match the oracle bit for bit: tableau, basis, status, pivots.
Usage: soak_dshard.py <seconds> [seed]   (progress -> stdout)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import torch
from yalps_amd import sharded
from tests import _oracle
o = _oracle.load(omp=True)
o.set_threads(8)
rng = np.random.default_rng(seed)
t_end, n, last, kinds = time.time() + budget, 0, time.time(), {}
while time.time() < t_end:
    small = rng.random() < 0.35  # forced onto shards with 1-3 rows per workgroup
    h = int(rng.integers(40, 1000)) if small else int(rng.integers(1025, 4500))
    w = int(rng.integers(4200, 16386)) if (small or h < 2600) and rng.random() < 0.6 else int(rng.integers(2100, 4200))
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > dens] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (1 if rng.random() < 0.6 else rng.choice([-1, 1], h - 1))
    if rng.random() < 0.3:
        m[1::7, 0] = 0.0  # degenerate rows: ratios <= precision
    m[0, 0] = 0.0
    m = m.reshape(-1)
    piv = float(rng.choice([1, 2, 3, 5, 9, 17, 40, 77]))
    os.environ["YALPS_HIP_DELAY_MIN_ROWS"] = "1" if small else "4"
    os.environ["YALPS_HIP_DELAY_DEPTH"] = str(rng.integers(2, 9))
    os.environ["YALPS_HIP_SHARD_NT"] = str(rng.integers(0, 2))
    ident = np.arange(w + h, dtype=np.int32)
    ref, rp, rv = m.copy(), ident.copy(), ident.copy()
    est, eres, epiv, _ = o.simplex(ref, w, h, rp, rv, max_pivots=piv)
    bounds = sharded.partition(h, 1)
    ops = sharded.HipShardOps(sharded.local_rows(m, w, h, bounds, 0), w, bounds, 0, h, ident, ident.copy(), device=0)
    st, res, np_ = sharded.sharded_simplex(ops, sharded.TorchComm(), max_pivots=piv, check_every=int(rng.integers(1, 9)))
    kernel = ops.tab.info()["streaming"]
    gm, gp, gv = ops.download()
    ops.close()
    kinds[kernel] = kinds.get(kernel, 0) + 1
    ok = (st, np_) == (est, epiv) and ((res != res and eres != eres) or res == eres) and \
        np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)
    if not ok:
        bad = np.argwhere(gm.reshape(h, w).view(np.int64) != ref.reshape(h, w).view(np.int64))
        print("MISMATCH", kernel, h, w, dens, piv, (st, np_, res), (est, epiv, eres), "bad cells", bad.shape[0], bad[:6].tolist(),
              dict((k, os.environ[k]) for k in ("YALPS_HIP_DELAY_MIN_ROWS", "YALPS_HIP_DELAY_DEPTH", "YALPS_HIP_SHARD_NT")), flush=True)
        sys.exit(1)
    n += 1
    if time.time() - last > 20:
        print("ok", n, "cases", flush=True); last = time.time()
print("soak passed:", n, "cases", kinds, flush=True)
