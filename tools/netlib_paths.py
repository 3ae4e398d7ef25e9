"""Which kernel each netlib problem's root LP runs on, and its time per pivot."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N, model as M, mps
from tests import _golden as G
ctx = N.Context(0)
for b in mps.read_benchmarks(os.path.join(G.GOLDEN, "netlib")):
    t0 = M.tableau_model(b["model"]).tableau
    t = N.DeviceTableau(ctx, t0.width, t0.height)
    best = None
    for _ in range(2):
        t.upload(t0.matrix, t0.height, t0.position_of_variable, t0.variable_at_position)
        st, res, piv, ms = t.solve(precision=b["options"]["precision"], max_pivots=math.inf, check_cycles=b["options"]["checkCycles"])
        best = ms if best is None else min(best, ms)
    info = t.info()
    kern = {"small": "small_kernel", "resident": info["resident"].split(" ")[0], "inplace": info["inplace"],
            "streaming": info["streaming"]}.get(info["last_path"], info["last_path"])
    print("%-10s %5dx%-5d %-28s %-10s pivots %6d  %8.3f ms  %6.2f us/pivot" % (b["name"], t0.height, t0.width, kern, st, piv, best, 1e3 * best / max(piv, 1)), flush=True)
    t.close()
