"""Per-call cost of the device-resident branch-and-cut node evaluation (apply_cuts / solve / downloads)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N, model as M
from tests import _cases as K
name = sys.argv[1] if len(sys.argv) > 1 else "Monster 2"
c = K.load(name)
tm = M.tableau_model(c["model"]); t = tm.tableau
ctx = N.Context(0)
root = N.DeviceTableau(ctx, t.width, t.height)
extra = 2 * len(tm.integers)
node = N.DeviceTableau(ctx, t.width, t.height + extra)
root.upload(t.matrix, t.height, t.position_of_variable, t.variable_at_position)
print("root", root.solve(max_pivots=1e9), root.info()["last_path"], t.height, t.width)
cuts = [(1, tm.integers[0], 0.0), (-1, tm.integers[1], 1.0)]
def tm_(f, n=200):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("apply_cuts us", tm_(lambda: node.apply_cuts(root, cuts)))
node.apply_cuts(root, cuts); print("node", node.solve(max_pivots=1e9), node.info()["last_path"])
def ev():
    node.apply_cuts(root, cuts); return node.solve(max_pivots=1e9)
print("apply+solve us", tm_(ev), ev()[2], "pivots")
print("download_rhs us", tm_(lambda: node.download_rhs()))
print("download perms us", tm_(lambda: node.download(matrix=False)))
print("download_solution us", tm_(lambda: node.download_solution()))
import ctypes as C
res, npiv = C.c_double(), C.c_int64()
lib = N.lib()
def bare():
    node.apply_cuts(root, cuts)
    return lib.yalps_tableau_solve(node.handle, 1e-8, 1e9, 0, C.byref(res), C.byref(npiv), None)
print("apply+solve (no timing events) us", tm_(bare))

def one_call():
    return node.node_solve(root, cuts, max_pivots=1e9)
os.environ["YALPS_HIP_NODE_FUSED"] = "0"
print("node_solve, call by call us", tm_(one_call), one_call()[0])
os.environ["YALPS_HIP_NODE_FUSED"] = "1"
print("node_solve, fused (3 launches) us", tm_(one_call), one_call()[0])
