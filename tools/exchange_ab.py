#!/usr/bin/env python3
"""The bare exchange of the register-resident kernels, flat against two-level (yalps_ctx_exchange_floor, persistent_floor.hip),
same box, alternating: the A/B behind DESIGN.md 4.2b's decision on the "XCD-leader" exchange.
  flat       every workgroup polls all NB 16-byte records
  two-level  the first workgroup of every XCD polls its XCD's records and raises ONE record per XCD; everybody polls those <= 8
each as: records only | + dependent fetch of the winner's 16 KB row | + every workgroup publishing a 16 KB row first.
usage (GPU box): python tools/exchange_ab.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
out = {"workgroups": 256, "lanes": 512, "row_bytes": 16384, "epochs": 4000, "rounds": []}
for rep in range(3):
    row = {}
    for name, variant in (("flat_records", 0), ("two_level_records", 4), ("flat_records_fetch", 2), ("two_level_records_fetch", 6),
                          ("flat_publish_records_fetch", 3), ("two_level_publish_records_fetch", 7)):
        nat.exchange_floor(ctx, 256, 512, 2, 200, variant)
        row[name] = round(nat.exchange_floor(ctx, 256, 512, 2, 4000, variant), 4)
    out["rounds"].append(row)
    print(row, flush=True)
out["best_us"] = {k: min(r[k] for r in out["rounds"]) for k in out["rounds"][0]}
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
ctx.close()
