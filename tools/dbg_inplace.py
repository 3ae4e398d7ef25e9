import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as nat
M, N, mp = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
w, h = N + 1, M + 1
m = nat.dense_lp(M, N, 11)
ctx = nat.Context(0)
t = nat.DeviceTableau(ctx, w, h)
ident = np.arange(w + h, dtype=np.int32)
t.upload(m, h, ident, ident.copy())
print("info", t.info(), flush=True)
print(t.solve(max_pivots=mp), t.info()["last_path"], flush=True)
