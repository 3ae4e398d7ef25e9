"""Randomised soak of the row-shard kernels with delayed row updates (dshard_kernel, dshard_select_kernel, dshard_sweep_kernel;
DESIGN.md 5) against the CPU oracle (test infrastructure, like tests/): ONE rank over RCCL -- the library's own loop,
yalps_shard_run, hipGraph replays included -- on seeded random tableaux of random shape, sparsity, signs of the right-hand sides
(phase-1 starts), pivot budget, delay depth, batch length, and the sweep inside the step kernel (through LDS panels or straight
from L2) or as a launch of its own; every solve must match the oracle bit for bit -- rows, basis, status, pivots.
Usage: soak_dshard_native.py <seconds> [seed]   (progress -> stdout)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # (torch's HIP runtime has to be initialised before the library's)
torch.cuda.init()
from yalps_amd import sharded
from tests import _oracle
o = _oracle.load(omp=True)
o.set_threads(8)
rng = np.random.default_rng(seed)
t_end, n, last, kinds = time.time() + budget, 0, time.time(), {}
while time.time() < t_end:
    wide = rng.random() < 0.35
    h = int(rng.integers(300, 7000))
    w = int(rng.integers(4100, 16385)) if wide and h < 2600 else int(rng.integers(300, 4200))
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > dens] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (1 if rng.random() < 0.6 else rng.choice([-1, 1], h - 1))
    if rng.random() < 0.3:
        m[1::7, 0] = 0.0  # degenerate rows: ratios <= precision
    m[0, 0] = 0.0
    m = m.reshape(-1)
    piv = float(rng.choice([1, 3, 9, 17, 40, 77, 131]))
    depth = int(rng.choice([2, 4, 8, 16, 5, 12]))
    every = int(rng.choice([16, 32, 8]))
    os.environ["YALPS_HIP_DELAY_MIN_ROWS"] = "1"
    os.environ["YALPS_HIP_DELAY_DEPTH"] = str(depth)
    for key, val in (("YALPS_HIP_SHARD_XSWEEP", rng.choice(["", "0", "1"])), ("YALPS_HIP_SHARD_PANEL", rng.choice(["", "0", "1"])),
                     ("YALPS_HIP_SHARD_NT", rng.choice(["", "0", "1"]))):
        if val:
            os.environ[key] = str(val)
        else:
            os.environ.pop(key, None)
    ident = np.arange(w + h, dtype=np.int32)
    ref, rp, rv = m.copy(), ident.copy(), ident.copy()
    est, eres, epiv, _ = o.simplex(ref, w, h, rp, rv, max_pivots=piv)
    bounds = sharded.partition(h, 1)
    ops = sharded.HipShardOps(m, w, bounds, 0, h, ident, ident.copy(), device=0, private_stream=True)
    comm = sharded.native_comm(ops.ctx, 0, 1, transport="rccl")
    st, res, np_, _ = ops.run_native(comm, max_pivots=piv, check_every=every)
    info = ops.tab.info()
    gm, gp, gv = ops.download()
    comm.close()
    ops.close()
    kind = info["streaming"] + "/" + info.get("shard_sweep", "?")
    kinds[kind] = kinds.get(kind, 0) + 1
    ok = (st, np_) == (est, epiv) and ((res != res and eres != eres) or res == eres) and \
        np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)
    if not ok:
        print("MISMATCH", info, h, w, dens, piv, depth, every, dict((k, os.environ.get(k)) for k in ("YALPS_HIP_SHARD_XSWEEP", "YALPS_HIP_SHARD_PANEL", "YALPS_HIP_SHARD_NT")),
              (st, np_, res), (est, epiv, eres), flush=True)
        sys.exit(1)
    n += 1
    if time.time() - last > 20:
        print("ok", n, "cases", flush=True); last = time.time()
print("soak passed:", n, "cases", kinds, flush=True)
