"""Randomised soak of the persistent kernels against the CPU oracle (test infrastructure, like tests/):
seeded random tableaux of random shape and sparsity, `pivots` pivots each, on the resident and the in-place
path; every solve must match the oracle bit for bit.  Usage: soak.py <seconds> [seed]   (progress -> stdout)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from yalps_amd import _native as N
from tests import _oracle
o = _oracle.load()
rng = np.random.default_rng(seed)
ctxs = {}
ctxs["small"] = N.Context(0)  # (only for tableaux that fit in LDS)
os.environ["YALPS_HIP_SMALL"] = "0"
os.environ["YALPS_HIP_RESIDENT"] = "1"; ctxs["resident"] = N.Context(0)
os.environ["YALPS_HIP_RESIDENT"] = "0"; ctxs["inplace"] = N.Context(0)
os.environ["YALPS_HIP_INPLACE"] = "0"; ctxs["streaming"] = N.Context(0)
t_end, n, last = time.time() + budget, 0, time.time()
LDS_VARIANTS = [(512, 1, 38), (512, 2, 16), (512, 3, 11), (512, 4, 7), (512, 5, 5), (512, 6, 3)]  # rows parked in LDS
n_lds = 0
while time.time() < t_end:
    tiny = rng.random() < 0.4
    h, w = (int(rng.integers(2, 120)), int(rng.integers(2, 120))) if tiny else (int(rng.integers(3, 1600)), int(rng.integers(3, 1600)))
    force = None
    if rng.random() < 0.2:  # tall and narrow, forced onto a resident variant with 1..8 rows per workgroup in LDS
        T, J, R = LDS_VARIANTS[int(rng.integers(len(LDS_VARIANTS)))]
        h = int(rng.integers(256 * R + 1, 256 * (R + 8) + 1))
        w = int(rng.integers(3, min(2 * T * J + 2, 40 if R > 16 else 160)))
        force = "%d,%d,%d" % (T, J, R)
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > dens] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (1 if rng.random() < 0.7 else rng.choice([-1, 1], h - 1))
    m[0, 0] = 0.0
    m = m.reshape(-1)
    piv = float(rng.choice([40, 150, 400]))
    chk = bool(rng.random() < 0.3)
    pos = np.arange(w + h, dtype=np.int32); var = pos.copy()
    ref, rp, rv = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = o.simplex(ref, w, h, rp, rv, max_pivots=piv, check_cycles=chk)
    for path, ctx in ctxs.items():
        if path == "small" and 8 * (h * (w + 4) + w + 4 * h) > 140 * 1024:
            continue
        if path == "streaming" and n % 8:
            continue  # (the launch-per-pivot kernels are slow: every 8th case)
        if force and path != "resident":
            continue
        if force:
            os.environ["YALPS_HIP_RVARIANT"] = force
        t = N.DeviceTableau(ctx, w, h)
        if force:
            del os.environ["YALPS_HIP_RVARIANT"]
            n_lds += 1
            assert t.info()["resident"] == "resident_kernel<%s,lds>" % force, t.info()
        t.upload(m, h, pos, var)
        st, res, np_, _ = t.solve(max_pivots=piv, check_cycles=chk)
        lp = t.info()["last_path"]
        gm, gp, gv = t.download()
        t.close()
        ok = (st, np_) == (est, epiv) and ((res != res and eres != eres) or res == eres) and \
            np.array_equal(gm.view(np.int64), ref.view(np.int64)) and np.array_equal(gp, rp) and np.array_equal(gv, rv)
        if not ok or lp != path:
            print("MISMATCH", path, lp, h, w, dens, piv, chk, (st, np_, res), (est, epiv, eres), flush=True)
            sys.exit(1)
    n += 1
    if time.time() - last > 20:
        print("ok", n, "cases", flush=True); last = time.time()
print("soak passed:", n, "cases (small where it fits, resident, in place, every 8th also launch per pivot),", n_lds,
      "of them tall ones on the resident variants with LDS rows", flush=True)
