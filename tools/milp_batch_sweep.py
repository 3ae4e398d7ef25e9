"""Whole-MILP time of the native driver (yalps_milp_f64) against the node-batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import solve as S
from tests import _cases as K
name = sys.argv[1] if len(sys.argv) > 1 else "Large Farm MIP"
c = K.load(name)
for nb in (0, 8, 32, 128, 512, 2048):
    st = {}
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); sol = S.solve(c["model"], c["options"], node_batch=nb, stats=st); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(name, "node_batch", nb, "%.1f ms" % (1e3 * best), sol["status"], sol["result"], st, flush=True)
