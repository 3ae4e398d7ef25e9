"""Writes tests/golden/c5_whole_solve.json: the whole solve of BASELINE config 5 (dense-LP(16384,16384,42), 83 270 pivots)
through stream3_kernel<512,16,nt> on one MI355X -- status, result, pivot count, M[0,0] and SHA-256 of column 0 and of both
permutations.  A HIP result (the oracle needs about an hour for it); tests/test_c5_parity.py pins the first 333 pivots of
the same run to the oracle and requires sweep_kernel and dshard_kernel to reproduce this record.
usage (GPU box): python tools/c5_record.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _c5  # noqa: E402
from tests.test_c5_parity import record_of  # noqa: E402
from yalps_amd import _native as nat  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "c5_whole_solve.json")
    ctx = nat.Context(0)
    t = nat.DeviceTableau(ctx, _c5.W, _c5.H)
    ident = np.arange(_c5.W + _c5.H, dtype=np.int32)
    t.upload(_c5.make_input(nat.dense_lp, "phase2"), _c5.H, ident, ident.copy())
    t0 = time.time()
    status, result, npiv, ms = t.solve(max_pivots=float("inf"))
    wall = time.time() - t0
    info = t.info()
    got, pos, var = t.download()
    got = got.reshape(_c5.H, _c5.W)
    rec = {"input": "dense-LP(16384,16384,seed=42), precision 1e-8, maxPivots Infinity, checkCycles false",
           "kernel": info["inplace"], "gpu_ms": ms, "wall_s": wall,
           "record": record_of(status, result, npiv, got[0, 0], got[:, 0], pos, var)}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
