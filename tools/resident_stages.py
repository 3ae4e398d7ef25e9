#!/usr/bin/env python3
"""Where a pivot of the resident kernel spends its time: in-kernel stage stamps (diagnostic build).

    python -c "from yalps_amd import build; build.build_hip(stamps=True)"      # here, once
    python tools/resident_stages.py [--size 2048] [--out profiles/r02_resident_stages.json]   # on the GPU box

Loads yalps_amd/libyalps_hip_stamps.so (compiled with -DYALPS_STAMPS: s_memtime sums per stage in scalar registers,
stored once per launch to a buffer nothing else reads; the shipped library executes no stamp), solves dense-LP(size,
size,42) and prints, per stage, the mean over workgroups of (cycles per pivot) and the same in microseconds at the
clock the launch held (s_memtime span / s_memrealtime span x 100 MHz).  Read the SHARES, not the length: the stamps'
waits forbid overlaps the real kernel has (CDNA guide 7, In-kernel stamps).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YALPS_HIP_LIB"] = os.path.join(ROOT, "yalps_amd", "libyalps_hip_stamps.so")

import numpy as np  # noqa: E402

from yalps_amd import _native  # noqa: E402

STAGES = {
    0: "wait for all flags (polling waves)", 1: "arg-min of the records (barrier)", 2: "fetch the winner's row",
    3: "phase 1 entering column / verdict", 4: "pivot column of my rows via LDS (barrier)",
    5: "normalise, column divisions, RHS, objective replica (barrier)", 6: "pricing (barrier)",
    7: "next entering column of my rows (barrier)", 8: "my candidate (barrier)", 9: "candidate row eliminated",
    10: "candidate row stores issued", 11: "rows eliminated while the stores drain", 12: "drain + barrier + flag store",
    13: "remaining rows", 14: "phase switch / unbounded exit",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--rows", type=int, default=0, help="M (default: size)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    N, M = args.size, args.rows or args.size
    w, h = N + 1, M + 1
    ctx = _native.Context(0)
    t = _native.DeviceTableau(ctx, w, h)
    m = _native.dense_lp(M, N, 42)
    ident = np.arange(w + h, dtype=np.int32)
    t.upload(m, h, ident, ident.copy())
    status, result, npiv, ms = t.solve(max_pivots=float("inf"))
    info = t.info()
    st = t.debug_stamps().astype(np.float64)
    t.close()
    ctx.close()
    piv = st[:, 20]
    clock_ghz = float(np.median(st[:, 21] / np.maximum(st[:, 22], 1.0)) * 0.1)
    per = st[:, :20] / piv[:, None]  # cycles per pivot, per workgroup
    out = {"workload": "dense-LP(%d,%d,42): tableau %dx%d" % (M, N, h, w), "kernel": info.get("resident"),
           "last_path": info.get("last_path"), "status": status, "pivots": int(npiv), "stamped_us_per_pivot": 1e3 * ms / npiv,
           "clock_ghz": clock_ghz, "workgroups": int(st.shape[0]),
           "note": "diagnostic build: read the shares; the stamps' waits forbid overlaps the real kernel has",
           "stages": []}
    tot = per.sum(axis=1).mean()
    for k in range(20):
        if per[:, k].max() == 0:
            continue
        c = per[:, k]
        out["stages"].append({"id": k, "name": STAGES.get(k, "?"), "cycles_mean": round(float(c.mean()), 1),
                              "us_mean": round(float(c.mean() / clock_ghz / 1e3), 3), "share": round(float(c.mean() / tot), 3),
                              "cycles_min_wg": round(float(c.min()), 1), "cycles_max_wg": round(float(c.max()), 1),
                              "wg0": round(float(c[0]), 1)})
    out["sum_us"] = round(float(tot / clock_ghz / 1e3), 3)
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
