"""One-rank row-sharded solves of wide tableaux against the oracle, under the shard kernels' switches (debugging aid)."""
import os, subprocess, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) <= 1:
    for shape in ("300 1500 2 8", "300 3000 2 8", "300 6000 2 8", "300 9000 2 8", "700 16384 4 1", "700 16384 4 2"):
        for env in ({}, {"YALPS_HIP_SHARD_INPLACE": "0"}):
            out = subprocess.run([sys.executable, __file__] + shape.split(), env=dict(os.environ, **env), capture_output=True, text=True)
            print(shape, env, out.stdout.strip()[-700:], out.stderr.strip()[-400:])
    sys.exit(0)
import torch
from tests import _oracle
from yalps_amd import sharded
M, N, seed, piv = (int(a) for a in sys.argv[1:5])
w, h = N + 1, M + 1
orc = _oracle.load()
m = orc.dense_lp(M, N, seed)
if seed % 2:  # "-a x <= -b": phase 1 first; exact zeros (untouched rows, flushed pivot-row entries); degenerate rows
    A = m.reshape(h, w)
    A[h // 3] *= -1.0
    A[5::7, 3::5] = 0.0
    A[2::9, 0] = 0.0
ident = np.arange(w + h, dtype=np.int32)
ref, rpos, rvar = m.copy(), ident.copy(), ident.copy()
est, eres, epiv, _ = orc.simplex(ref, w, h, rpos, rvar, max_pivots=float(piv))
bounds = sharded.partition(h, 1)
ops = sharded.HipShardOps(sharded.local_rows(m, w, h, bounds, 0), w, bounds, 0, h, ident, ident.copy(), device=0)
status, result, pivots = sharded.sharded_simplex(ops, sharded.TorchComm(), max_pivots=float(piv), check_every=1)
lm, pos, var = ops.download()
info = ops.tab.info()
got = lm.reshape(h, w); exp = ref.reshape(h, w)
bad = np.argwhere(got.view(np.int64) != exp.view(np.int64))
rows = sorted(set(bad[:, 0].tolist()))
print(json.dumps({"shape": [h, w], "oracle": [est, epiv], "gpu": [status, pivots], "streaming": info["streaming"], "bad_cells": int(bad.shape[0]),
                  "bad_rows": rows[:10], "n_bad_rows": len(rows), "first": bad[:6].tolist(),
                  "pos_ok": bool(np.array_equal(pos, rpos)), "var_ok": bool(np.array_equal(var, rvar))}))
