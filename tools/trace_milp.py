"""One native branch-and-cut solve of a reference MILP case (default Monster 2) after a warm-up, for a rocprofv3
--hip-trace --kernel-trace timeline of the per-node cost on the device-node evaluator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _cases as K
from yalps_amd import solve as S
name = sys.argv[1] if len(sys.argv) > 1 else "Monster 2"
case = K.load(name)
for rep in range(3):
    stats = {}
    t0 = time.perf_counter()
    sol = S.solve(case["model"], case["options"], stats=stats)
    print(name, sol["status"], sol["result"], "%.2f ms" % ((time.perf_counter() - t0) * 1e3), stats, flush=True)
