"""Randomised soak of the delayed-update kernels (stream3_kernel / stream2_kernel) against the CPU oracle (test
infrastructure, like tests/): seeded random tableaux of random shape, sparsity, signs of the right-hand sides (phase-1 starts),
pivot budget, delay depth and kernel; every solve must match the oracle bit for bit -- tableau, basis, status, pivots.
Usage: soak_delay.py <seconds> [seed]   (progress -> stdout)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ["YALPS_HIP_SMALL"] = "0"
os.environ["YALPS_HIP_RESIDENT"] = "0"
from yalps_amd import _native as N
from tests import _oracle
o = _oracle.load(omp=True)
o.set_threads(8)
ctx = N.Context(0)
rng = np.random.default_rng(seed)
t_end, n, last, kinds = time.time() + budget, 0, time.time(), {}
while time.time() < t_end:
    wide = rng.random() < 0.3
    h = int(rng.integers(1025, 5000))
    w = int(rng.integers(2049, 16385)) if wide and h < 2600 else int(rng.integers(3, 4200))
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > dens] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (1 if rng.random() < 0.6 else rng.choice([-1, 1], h - 1))
    if rng.random() < 0.3:
        m[1::7, 0] = 0.0  # degenerate rows: ratios <= precision
    m[0, 0] = 0.0
    m = m.reshape(-1)
    piv = float(rng.choice([1, 2, 3, 5, 9, 17, 40, 77, 131]))
    os.environ["YALPS_HIP_DELAY_KERNEL"] = str(rng.choice([2, 3, 3]))
    os.environ["YALPS_HIP_DELAY_DEPTH"] = str(rng.integers(2, 17))  # (round 3: up to 16 pending pivots)
    panel = rng.choice(["", "0", "1"])  # the sweep through LDS panels / straight from L2: by rows per workgroup, or forced
    if panel:
        os.environ["YALPS_HIP_STREAM3_PANEL"] = str(panel)
    else:
        os.environ.pop("YALPS_HIP_STREAM3_PANEL", None)
    os.environ["YALPS_HIP_DELAY_NT"] = str(rng.integers(0, 2))
    pos = np.arange(w + h, dtype=np.int32); var = pos.copy()
    ref, rp, rv = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = o.simplex(ref, w, h, rp, rv, max_pivots=piv)
    t = N.DeviceTableau(ctx, w, h)
    t.upload(m, h, pos, var)
    st, res, np_, _ = t.solve(max_pivots=piv)
    info = t.info()
    gm, gp, gv = t.download()
    t.close()
    kinds[info["inplace"] + "/" + info.get("sweep", "?")] = kinds.get(info["inplace"] + "/" + info.get("sweep", "?"), 0) + 1
    ok = (st, np_) == (est, epiv) and ((res != res and eres != eres) or res == eres) and \
        np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)
    if not ok or info["last_path"] != "inplace":  # (stream2 where asked for and built, else stream3; sweep / stream_kernel below four rows per workgroup)
        print("MISMATCH", info, h, w, dens, piv, (st, np_, res), (est, epiv, eres), flush=True)
        sys.exit(1)
    n += 1
    if time.time() - last > 20:
        print("ok", n, "cases", flush=True); last = time.time()
print("soak passed:", n, "cases", kinds, flush=True)
