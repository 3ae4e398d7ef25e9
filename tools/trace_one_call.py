"""A few device solves of 1 pivot each on the resident path, for a rocprofv3 --hip-trace --kernel-trace
--memory-copy-trace timeline of the fixed per-call cost (see tools/resident_call_overhead.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YALPS_HIP_SMALL"] = "0"
from yalps_amd import _native as N
ctx = N.Context(0)
M = 512
w = h = M + 1
m = N.dense_lp(M, M, 42)
pos = np.arange(w + h, dtype=np.int32)
t = N.DeviceTableau(ctx, w, h)
for rep in range(6):
    t.upload(m, h, pos, pos.copy())
    t.solve(max_pivots=1.0)
t.close()
