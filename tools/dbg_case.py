"""Replays one case of tests/test_hip_parity.py::test_degenerate_integer_lps_on_every_path against the oracle under a few
kernel selections (debugging aid)."""
import os, subprocess, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
want = int(sys.argv[1]) if len(sys.argv) > 1 else 72
if len(sys.argv) <= 2:
    for env in ({}, {"YALPS_HIP_RESIDENT_GEN": "1"}, {"YALPS_HIP_TAG": "0"}, {"YALPS_HIP_TAG": "0", "YALPS_HIP_RESIDENT_GEN": "1"}, {"YALPS_HIP_RESIDENT": "0"}):
        out = subprocess.run([sys.executable, __file__, str(want), "child"], env=dict(os.environ, YALPS_HIP_SMALL="0", **env), capture_output=True, text=True)
        print(env, out.stdout.strip()[-600:], out.stderr.strip()[-300:])
    sys.exit(0)
from yalps_amd import _native as nat
from tests import _oracle
oracle = _oracle.load()
rng = np.random.default_rng(20240607)
for case in range(84):
    big = case >= 70
    h, w = (int(rng.integers(300, 1200)), int(rng.integers(40, 500))) if big else (int(rng.integers(2, 70)), int(rng.integers(2, 90)))
    m = rng.integers(-3, 4, size=(h, w)).astype(np.float64)
    m[rng.random((h, w)) < rng.choice([0.0, 0.3, 0.7])] = 0.0
    m[1:, 0] = rng.integers(-1 if case % 3 == 0 else 0, 5, size=h - 1)
    m[0, 0] = 0.0
    if case % 5 == 0:
        m *= 0.5
    m = m.reshape(-1)
    opts = dict(precision=float(rng.choice([1e-8, 1e-6, 1e-12])), max_pivots=float(rng.choice([3000, 7, 60])), check_cycles=bool(rng.integers(0, 2)))
    if case != want:
        continue
    pos, var = np.arange(w + h, dtype=np.int32), np.arange(w + h, dtype=np.int32)
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, trace = oracle.simplex(ref, w, h, rpos, rvar, **opts)
    c = nat.Context(0)
    t = nat.DeviceTableau(c, w, h)
    t.upload(m, h, pos, var)
    status, result, npiv, _ = t.solve(**opts)
    got, gpos, gvar = t.download()
    info = t.info()
    bad_pos = np.nonzero(gpos != rpos)[0]
    bad_var = np.nonzero(gvar != rvar)[0]
    bad_m = np.nonzero(got.view(np.int64) != ref.view(np.int64))[0]
    print(json.dumps({"case": case, "h": h, "w": w, "opts": opts, "oracle": [est, eres, epiv], "gpu": [status, result, npiv],
                      "kernel": info["resident"], "path": info["last_path"], "launches": info["last_resident_launches"],
                      "bad_pos": bad_pos[:8].tolist(), "gpos": gpos[bad_pos[:8]].tolist(), "rpos": rpos[bad_pos[:8]].tolist(),
                      "bad_var": bad_var[:8].tolist(), "gvar": gvar[bad_var[:8]].tolist(), "rvar": rvar[bad_var[:8]].tolist(),
                      "bad_matrix_cells": int(bad_m.size)}))
