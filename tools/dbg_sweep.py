import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as nat
from tests import _np_simplex as NP
M, N, pivots = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
plain = len(sys.argv) > 4 and sys.argv[4] == "plain"
w, h = N + 1, M + 1
m = nat.dense_lp(M, N, 17)
A = m.reshape(h, w)
if not plain:
    A[h // 3] *= -1.0
    A[5::7, 3::5] = 0.0
    A[2::9, 0] = 0.0
pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
est, eres, epiv = NP.simplex(ref, w, h, rpos, rvar, max_pivots=pivots)
ctx = nat.Context(0)
t = nat.DeviceTableau(ctx, w, h)
t.upload(m, h, pos, var)
status, result, npiv, _ = t.solve(max_pivots=pivots)
print(t.info())
got, gpos, gvar = t.download()
print(status, npiv, result, "| expected", est, epiv, eres)
d = (got.view(np.int64) != ref.view(np.int64)).reshape(h, w)
print("differing cells", d.sum(), "rows", np.flatnonzero(d.any(1))[:20], "cols", np.flatnonzero(d.any(0))[:20], "n cols", d.any(0).sum(), "n rows", d.any(1).sum())
rr, cc = np.nonzero(d)
for r, c in list(zip(rr, cc))[:8]:
    g, e = got.reshape(h, w)[r, c], ref.reshape(h, w)[r, c]
    print(r, c, g, e, hex(np.float64(g).view(np.uint64)), hex(np.float64(e).view(np.uint64)), "orig", hex(np.float64(m.reshape(h, w)[r, c]).view(np.uint64)), m.reshape(h,w)[r,c])
# is the wrong value explained by a wrong pivot-row entry?  x_new = x - coef * p  =>  p_used = (x - x_new) / coef
if len(sys.argv) > 3 and int(sys.argv[3]) == 1 and len(rr):
    o = m.reshape(h, w)
    r, c = rr[0], cc[0]
    print("row of first diff", r, "cells differing in that row:", int(d[r].sum()))
print("perm ok", np.array_equal(gpos, rpos), np.array_equal(gvar, rvar))
