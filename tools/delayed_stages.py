#!/usr/bin/env python3
"""Where a pivot of the delayed-update kernels spends its time: in-kernel stage stamps (diagnostic build).

    python -c "from yalps_amd import build; build.build_hip(stamps=True)"                 # here, once
    python tools/delayed_stages.py --kernel stream3 --size 16384 [--rows R] [--pivots 400] [--out profiles/...json]
    python tools/delayed_stages.py --kernel dshard  --size 16384 --rows 2048              # one rank's share of 8

Loads yalps_amd/libyalps_hip_stamps.so (-DYALPS_STAMPS: s_memtime sums per stage in scalar registers, added once per
launch to a buffer nothing else reads; the shipped library executes no stamp).  stream3: the persistent in-place kernel
(yalps_tableau_solve); dshard: the row-shard step kernel, one rank, the Python loop (select kernel, copy, step kernel
per pivot -- only the step kernel is stamped; its launches are summed).  Per stage: mean over workgroups of cycles per
pivot, the same in microseconds at the clock the launches held, the slowest workgroup and workgroup 0.  Read the
SHARES: the stamps' waits forbid overlaps the real kernel has.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YALPS_HIP_LIB"] = os.path.join(ROOT, "yalps_amd", "libyalps_hip_stamps.so")

import numpy as np  # noqa: E402

if "dshard" in sys.argv:  # (the shard driver's slots are torch tensors: torch's HIP runtime has to be initialised first)
    import torch  # noqa: E402
    torch.cuda.init()
from yalps_amd import _native  # noqa: E402

STAGES = {
    "stream3": {
        0: "everybody's key record polled, arg-min (barrier)",
        1: "two-step exchange: owner runs the winner's row through the pending pivots + publishes it / others wait",
        2: "phase 1 entering column / checkCycles verdict",
        3: "my rows' pivot-column entries, quotient (two barriers)",
        4: "RHS entries, what replaces the pivot column",
        5: "raw pivot row (sc1) -> normalised -> scratch, objective replica, priced; stores drained",
        6: "arg-max of the pricing (barriers)",
        7: "my rows' entries of the next entering column (scalar chains)",
        8: "my candidate (arg-min) + key record",
        9: "basis bookkeeping (workgroup 0)",
        10: "the sweep (every depth-th pivot) / barrier",
        18: "sweep: barrier in front of the fill (the other waves' trips)", 11: "sweep: panel fill (L2 -> LDS)", 12: "sweep: barrier behind the fill / between trips", 13: "sweep: my rows' loads (waited for here in this build)",
        14: "sweep: the pending pivots applied in registers", 15: "sweep: stores issued", 16: "sweep: tail units", 17: "sweep: last barrier",
    },
    "dshard": {
        0: "state + the pending pivots' scalars of my rows -> LDS (launch prologue)",
        1: "decide: the gathered records (phase 1: entering column)",
        2: "my rows' entries of the pivot column",
        3: "RHS, what replaces the pivot column; pending scalars stored",
        4: "pivot row: normalised + stored, objective replica, priced in registers",
        5: "arg-max of the pricing (barriers)",
        6: "my rows' entries of the next entering column (scalar chains)",
        7: "my candidates, two arg-mins, the partials",
        8: "state",
        9: "the sweep (every depth-th launch)",
        18: "sweep: barrier in front of the fill (the other waves' trips)", 11: "sweep: panel fill (L2 -> LDS)", 12: "sweep: barrier behind the fill / between trips", 13: "sweep: my rows' loads (waited for here in this build)",
        14: "sweep: the pending pivots applied in registers", 15: "sweep: stores issued", 16: "sweep: tail units", 17: "sweep: last barrier",
    },
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", choices=("stream3", "dshard"), default="stream3")
    ap.add_argument("--size", type=int, default=16384, help="N: columns of dense-LP(M,N,42)")
    ap.add_argument("--rows", type=int, default=0, help="M (default: size)")
    ap.add_argument("--pivots", type=int, default=400)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    N, M = args.size, args.rows or args.size
    w, h = N + 1, M + 1
    m = _native.dense_lp(M, N, 42)
    ident = np.arange(w + h, dtype=np.int32)
    if args.kernel == "stream3":
        ctx = _native.Context(0)
        t = _native.DeviceTableau(ctx, w, h)
        t.upload(m, h, ident, ident.copy())
        status, result, npiv, ms = t.solve(max_pivots=float(args.pivots))
        info = t.info()
        kernel = info.get("inplace")
        st = t.debug_stamps().astype(np.float64)
        t.close()
        ctx.close()
        wall_us = 1e3 * ms / max(npiv, 1)
    else:
        import time
        from yalps_amd import sharded
        bounds = sharded.partition(h, 1)
        ops = sharded.HipShardOps(m, w, bounds, 0, h, ident, ident.copy(), device=0)
        ops.tab.debug_stamps()  # reset
        t0 = time.perf_counter()
        status, result, npiv = sharded.sharded_simplex(ops, sharded.TorchComm(), max_pivots=float(args.pivots), check_every=64)
        wall_us = 1e6 * (time.perf_counter() - t0) / max(npiv, 1)
        kernel = ops.tab.info()["streaming"]
        st = ops.tab.debug_stamps().astype(np.float64)
        ops.close()
    piv = np.maximum(st[:, 20], 1.0)
    clock_ghz = float(np.median(st[:, 21] / np.maximum(st[:, 22], 1.0)) * 0.1)
    per = st[:, :20] / piv[:, None]
    names = STAGES[args.kernel]
    out = {"workload": "dense-LP(%d,%d,42): tableau %dx%d, %d pivots" % (M, N, h, w, npiv), "kernel": kernel, "status": status,
           "pivots": int(npiv), "stamped_us_per_pivot": wall_us, "clock_ghz": clock_ghz, "workgroups": int(st.shape[0]),
           "stamped_launches_or_pivots_per_workgroup": float(piv.mean()),
           "note": "diagnostic build: read the shares; the stamps' waits forbid overlaps the real kernel has"
                   + ("; dshard: wall time includes the select kernel, the copy and the Python loop, the stages only the step kernel" if args.kernel == "dshard" else ""),
           "stages": []}
    tot = per.sum(axis=1).mean()
    for k in range(20):
        if per[:, k].max() == 0:
            continue
        c = per[:, k]
        out["stages"].append({"id": k, "name": names.get(k, "?"), "us_mean": round(float(c.mean() / clock_ghz / 1e3), 3),
                              "share": round(float(c.mean() / tot), 3), "us_min_wg": round(float(c.min() / clock_ghz / 1e3), 3),
                              "us_max_wg": round(float(c.max() / clock_ghz / 1e3), 3), "us_wg0": round(float(c[0] / clock_ghz / 1e3), 3)})
    out["sum_us"] = round(float(tot / clock_ghz / 1e3), 3)
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
