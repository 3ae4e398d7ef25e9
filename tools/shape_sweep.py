"""Whole-solve rate of dense-LP(M,N,seed=42) over a grid of shapes: which kernel runs, us per pivot,
algorithmic TB/s (16*h*w per pivot).  For DESIGN.md's path-selection table."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [
    (32, 32), (64, 128), (128, 128), (256, 256), (512, 512), (1024, 1024), (1536, 1536), (2048, 2048), (2560, 2560),
    (3072, 3072), (3300, 3000), (11000, 900), (4096, 4096), (512, 4096), (4096, 512), (256, 8192), (1024, 16384)]
ctx = N.Context(0)
rows = []
for M, Nn in shapes:
    w, h = Nn + 1, M + 1
    m = N.dense_lp(M, Nn, 42)
    ident = np.arange(w + h, dtype=np.int32)
    t = N.DeviceTableau(ctx, w, h)
    best = None
    for rep in range(2):
        t.upload(m, h, ident, ident.copy())
        st, res, piv, ms = t.solve(max_pivots=float("inf"))
        best = ms if best is None else min(best, ms)
    info = t.info()
    kern = {"small": "small_kernel", "resident": info["resident"].split(" ")[0], "inplace": info["inplace"],
            "streaming": info["streaming"]}.get(info["last_path"], info["last_path"])
    us = 1e3 * best / max(piv, 1)
    rows.append({"tableau": "%dx%d" % (h, w), "kernel": kern, "status": st, "pivots": piv, "ms": round(best, 3),
                 "us_per_pivot": round(us, 2), "algorithmic_TBps": round(16.0 * h * w / us / 1e6, 3)})
    print(rows[-1], flush=True)
    t.close()
print(json.dumps({"rows": rows}))
