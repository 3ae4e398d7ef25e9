"""Fixed cost of one device solve on the resident path (tableau already in HBM): wall time of solve(max_pivots=k) for
k = 1 and k = 201 on a 513x513 dense LP -> per-call overhead and marginal time per pivot."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YALPS_HIP_SMALL"] = "0"
from yalps_amd import _native as N
ctx = N.Context(0)
for M in (200, 512, 1024, 2048):
    w = h = M + 1
    m = N.dense_lp(M, M, 42)
    pos = np.arange(w + h, dtype=np.int32)
    t = N.DeviceTableau(ctx, w, h)
    out = {}
    for k in (1, 101):
        best = 1e9
        for rep in range(20):
            t.upload(m, h, pos, pos.copy())
            t0 = time.perf_counter()
            st, res, piv, ms = t.solve(max_pivots=float(k))
            best = min(best, (time.perf_counter() - t0) * 1e6)
        out[k] = best
    per = (out[101] - out[1]) / 100.0
    print("%dx%d %s: call with 1 pivot %.1f us, with 101 pivots %.1f us -> %.2f us/pivot marginal, fixed %.1f us"
          % (h, w, t.info()["resident"], out[1], out[101], per, out[1] - per), flush=True)
    t.close()
