#!/usr/bin/env python3
"""DESIGN.md 4.8, the experiment: stream_kernel<1024,8> -- the instantiation that "needs 77 VGPR + 118 SGPR spills and computed
garbage on the GPU" in round 1 and was dropped -- rebuilt from today's sources (tools/build_variant.sh stream8
-DYALPS_EXPERIMENT_STREAM8 stream: 65 VGPR + 117 SGPR spills, 228 bytes of scratch per lane) and run against the oracle.
If it is bit-exact now, scratch use as such is not what produced the wrong rows then.  GPU box; prints one line per shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YALPS_HIP_LIB"] = os.path.join(ROOT, "yalps_amd", "libyalps_hip_stream8.so")
os.environ.update(YALPS_HIP_SWEEP="0", YALPS_HIP_DELAY="0", YALPS_HIP_RESIDENT="0", YALPS_HIP_SMALL="0")

import numpy as np  # noqa: E402

from tests import _oracle  # noqa: E402
from yalps_amd import _native as nat  # noqa: E402

orc = _oracle.load(omp=True)
orc.set_threads(8)
ctx = nat.Context(0)
bad_total = 0
for M, N, pivots in ((600, 16000, 70), (2100, 12345, 50), (300, 9000, 60), (1025, 16384, 120), (4000, 8200, 40)):
    w, h = N + 1, M + 1
    m = nat.dense_lp(M, N, 17)
    A = m.reshape(h, w)
    A[h // 3] *= -1.0
    A[5::7, 3::5] = 0.0
    A[2::9, 0] = 0.0
    pos = np.arange(w + h, dtype=np.int32)
    var = pos.copy()
    ref, rpos, rvar = m.copy(), pos.copy(), var.copy()
    est, eres, epiv, _ = orc.simplex(ref, w, h, rpos, rvar, max_pivots=float(pivots))
    t = nat.DeviceTableau(ctx, w, h)
    t.upload(m, h, pos, var)
    status, result, npiv, _ = t.solve(max_pivots=float(pivots))
    info = t.info()
    got, gpos, gvar = t.download()
    t.close()
    wrong_rows = int((got.reshape(h, w).view(np.int64) != ref.reshape(h, w).view(np.int64)).any(axis=1).sum())
    ok = (status, npiv) == (est, epiv) and wrong_rows == 0 and np.array_equal(gpos, rpos) and np.array_equal(gvar, rvar)
    bad_total += 0 if ok else 1
    print("%5d x %5d  %3d pivots  path=%s kernel=%s  status %s/%s pivots %d/%d  rows that differ: %d  -> %s"
          % (h, w, pivots, info["last_path"], info["inplace"], status, est, npiv, epiv, wrong_rows, "bit-exact" if ok else "WRONG"), flush=True)
ctx.close()
print("stream_kernel<1024,8> with 228 B of scratch per lane: %s" % ("bit-exact on every shape" if bad_total == 0 else "%d shapes wrong" % bad_total))
