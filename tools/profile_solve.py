#!/usr/bin/env python3
"""A bounded solve for the profiler: dense-LP(M,N,42) for at most --pivots pivots, one JSON line (kernel, pivots, HIP-event
time of the pivot loop, us per pivot, algorithmic TB/s).  rocprofv3 --kernel-trace --stats / --pmc wrap this command.
  python3 tools/profile_solve.py --size 16384 --pivots 300 [--rows M]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, required=True)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--pivots", type=float, default=float("inf"))
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
Nn, M = a.size, a.rows or a.size
w, h = Nn + 1, M + 1
ctx = N.Context(0)
t = N.DeviceTableau(ctx, w, h)
m = N.dense_lp(M, Nn, 42)
ident = np.arange(w + h, dtype=np.int32)
best = None
for _ in range(a.reps):
    t.upload(m, h, ident, ident.copy())
    st, res, piv, ms = t.solve(max_pivots=a.pivots)
    best = ms if best is None else min(best, ms)
info = t.info()
kern = {"small": "small_kernel", "resident": info["resident"].split(" ")[0], "inplace": info["inplace"],
        "streaming": info["streaming"]}.get(info["last_path"], info["last_path"])
bpp = 16 * (h - 1) * w + 16 * w + 8 * (h - 1) + 8 * (w - 1) + 16 * (h - 1)
print(json.dumps({"tableau": "%dx%d" % (h, w), "kernel": kern, "status": st, "pivots": piv, "launches": int(info["last_resident_launches"]),
                  "ms": best, "us_per_pivot": 1e3 * best / max(piv, 1), "algorithmic_bytes_per_pivot": bpp,
                  "algorithmic_TBps": bpp * piv / (best * 1e-3) / 1e12}))
t.close()
ctx.close()
