import sys, threading, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N
M = 1536
w = h = M + 1
m = N.dense_lp(M, M, 42)
pos = np.arange(w + h, dtype=np.int32)
out = {}
def work(k):
    c = N.Context(0)
    t = N.DeviceTableau(c, w, h)
    res = []
    for rep in range(6):
        t.upload(m, h, pos, pos.copy())
        st, r, piv, ms = t.solve(max_pivots=float("inf"))
        res.append((st, r, piv, round(ms, 2), t.info()["last_path"]))
    out[k] = res
    t.close(); c.close()
ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[x.start() for x in ths]; [x.join() for x in ths]
for k in out: print(k, out[k])
