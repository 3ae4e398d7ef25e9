"""Apply-only bandwidth of the streaming kernels: back-to-back launches of one Gauss-Jordan sweep
(yalps_tableau_bench_sweep) on dense tableaux of several shapes.  Prints us per launch and
algorithmic TB/s (16*h*w bytes per launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as N

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(2049, 2049), (4097, 4097), (8193, 8193), (2049, 16385)]
ctx = N.Context(0)
for h, w in shapes:
    rng = np.random.default_rng(1)
    m = rng.uniform(0.5, 1.5, h * w)
    t = N.DeviceTableau(ctx, w, h)
    ident = np.arange(w + h, dtype=np.int32)
    t.upload(m, h, ident, ident.copy())
    del m
    t.bench_sweep(h // 2, w // 2, 5)
    us = t.bench_sweep(h // 2, w // 2, 40)
    print("%dx%d %s: %.1f us/launch, %.2f TB/s" % (h, w, t.info().get("streaming"), us, 16.0 * h * w / us / 1e6), flush=True)
    t.close()
