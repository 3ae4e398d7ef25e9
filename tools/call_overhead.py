import time, sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from yalps_amd import _native as N, model as M
from tests import _cases as K
c = K.load("Large Farm MIP")
tm = M.tableau_model(c["model"]); t = tm.tableau
w, h = t.width, t.height
ctx = N.Context(0)
dt = N.DeviceTableau(ctx, w, h)
def tm_(f, n=300):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
m, pos, var = t.matrix.copy(), t.position_of_variable.copy(), t.variable_at_position.copy()
print("upload us", tm_(lambda: dt.upload(m, h, pos, var)))
dt.upload(m, h, pos, var); print(dt.solve(max_pivots=1e9))
print("solve(0 pivots) us", tm_(lambda: dt.solve(max_pivots=1e9)))
print("download full us", tm_(lambda: dt.download()))
print("download rhs us", tm_(lambda: dt.download_rhs()))
def full():
    mm, p, v = t.matrix.copy(), pos.copy(), var.copy()
    N.simplex_host(mm, w, h, p, v, max_pivots=1e9)
print("drop-in 34 pivots us", tm_(full))
mm, p, v = t.matrix.copy(), pos.copy(), var.copy()
N.simplex_host(mm, w, h, p, v, max_pivots=1e9)
print("drop-in 0 pivots us", tm_(lambda: N.simplex_host(mm, w, h, p, v, max_pivots=1e9)))
def up_solve():
    dt.upload(m, h, pos, var); return dt.solve(max_pivots=1e9)
print("upload+solve 34 pivots us", tm_(up_solve), up_solve())
