"""Device solves (tableau already in HBM) of small dense LPs through small_kernel (one workgroup, LDS) and through the
resident kernel: wall time of a call with 1 and with 41 pivots -> fixed cost and marginal time per pivot of each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N
for small in ("1", "0"):
    os.environ["YALPS_HIP_SMALL"] = small
    ctx = N.Context(0)
    for M, Nn in ((32, 32), (64, 64), (96, 96), (128, 128), (150, 104), (36, 100), (176, 80)):
        w, h = Nn + 1, M + 1
        m = N.dense_lp(M, Nn, 42)
        pos = np.arange(w + h, dtype=np.int32)
        t = N.DeviceTableau(ctx, w, h)
        out = {}
        for k in (1, 41):
            best = 1e9
            for rep in range(20):
                t.upload(m, h, pos, pos.copy())
                t0 = time.perf_counter()
                st, res, piv, ms = t.solve(max_pivots=float(k))
                best = min(best, (time.perf_counter() - t0) * 1e6)
            out[k] = (best, piv)
        per = (out[41][0] - out[1][0]) / max(out[41][1] - out[1][1], 1)
        print("SMALL=%s %dx%d %-10s 1 pivot %.1f us, %d pivots %.1f us -> %.2f us/pivot, fixed %.1f us"
              % (small, h, w, t.info()["last_path"], out[1][0], out[41][1], out[41][0], per, out[1][0] - per), flush=True)
        t.close()
    ctx.close()
