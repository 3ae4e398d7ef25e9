"""N branch-and-cut nodes of one problem through yalps_tableau_node_solve, for a rocprofv3 --kernel-trace
--memory-copy-trace timeline of one node's life (YALPS_HIP_NODE_FUSED=0/1 chooses call by call / three launches).
  python3 tools/trace_node.py "Monster 2" 30"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N, model as M
from tests import _cases as K
name = sys.argv[1] if len(sys.argv) > 1 else "Monster 2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
c = K.load(name)
tm = M.tableau_model(c["model"]); t = tm.tableau
ctx = N.Context(0)
root = N.DeviceTableau(ctx, t.width, t.height)
node = N.DeviceTableau(ctx, t.width, t.height + 2 * len(tm.integers))
root.upload(t.matrix, t.height, t.position_of_variable, t.variable_at_position)
root.solve(max_pivots=1e9)
cuts = [(1, tm.integers[0], 0.0), (-1, tm.integers[1], 1.0)]
node.node_solve(root, cuts, max_pivots=1e9)
t0 = time.perf_counter()
for _ in range(n):
    out = node.node_solve(root, cuts, max_pivots=1e9)
print(name, out[0], "fused=" + os.environ.get("YALPS_HIP_NODE_FUSED", "1"), "us per node", (time.perf_counter() - t0) / n * 1e6, node.info())
