"""Replays tools/soak_delay.py's random sequence (same seed) up to the case of a given shape and shows where the GPU result
differs from the oracle (debugging aid).  dbg_soak_case.py SEED H W [env overrides KEY=VAL ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seed, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
over = dict(a.split("=") for a in sys.argv[4:])
os.environ["YALPS_HIP_SMALL"] = "0"
os.environ["YALPS_HIP_RESIDENT"] = "0"
from yalps_amd import _native as N
from tests import _oracle
o = _oracle.load(omp=True)
o.set_threads(8)
rng = np.random.default_rng(seed)
for case in range(100000):
    wide = rng.random() < 0.3
    h = int(rng.integers(1025, 5000))
    w = int(rng.integers(2049, 16385)) if wide and h < 2600 else int(rng.integers(3, 4200))
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    hit = (h, w) == (H, W)
    m = rng.uniform(-1, 1, (h, w))
    m[rng.random((h, w)) > dens] = 0.0
    m[1:, 0] = np.abs(m[1:, 0]) * (1 if rng.random() < 0.6 else rng.choice([-1, 1], h - 1))
    if rng.random() < 0.3:
        m[1::7, 0] = 0.0
    m[0, 0] = 0.0
    m = m.reshape(-1)
    piv = float(rng.choice([1, 2, 3, 5, 9, 17, 40, 77]))
    env = {"YALPS_HIP_DELAY_KERNEL": str(rng.choice([2, 3, 3])), "YALPS_HIP_DELAY_DEPTH": str(rng.integers(2, 9)), "YALPS_HIP_DELAY_NT": str(rng.integers(0, 2))}
    if not hit:
        continue
    env.update(over)
    os.environ.update(env)
    print("case", case, h, w, dens, piv, env, flush=True)
    ctx = N.Context(0)
    for budget in ([piv] if "ALL" not in over else list(range(int(over.get("FROM", 1)), int(piv) + 1))):
        pos = np.arange(w + h, dtype=np.int32); var = pos.copy()
        ref, rp, rv = m.copy(), pos.copy(), var.copy()
        est, eres, epiv, trace = o.simplex(ref, w, h, rp, rv, max_pivots=float(budget), trace_cap=128)
        t = N.DeviceTableau(ctx, w, h)
        t.upload(m, h, pos, var)
        st, res, np_, _ = t.solve(max_pivots=float(budget))
        info = t.info()
        gm, gp, gv = t.download()
        t.close()
        bad = np.argwhere(gm.reshape(h, w).view(np.int64) != ref.reshape(h, w).view(np.int64))
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print(budget, info["inplace"], info.get("delay_depth"), (st, np_), (est, epiv), "bad cells", len(bad), "rows", rows[:8], len(rows), "cols", cols[:8], len(cols),
              "pos ok", bool(np.array_equal(gp, rp)), "var ok", bool(np.array_equal(gv, rv)), flush=True)
        if len(bad):
            r, c = bad[0]
            print("   first", (int(r), int(c)), gm.reshape(h, w)[r, c], ref.reshape(h, w)[r, c], "last pivots (row, col)", trace[-10:].tolist())
            print("   bad cols range", cols[0], cols[-1], "bad rows sample", rows[:20])
            # which pivots had their pivot column / row in the bad set
            print("   pivots whose column is a bad column:", [(k + 1, tuple(x)) for k, x in enumerate(trace.tolist()) if x[1] in set(cols)][-5:])
            if "ALL" in over:
                break
    break
