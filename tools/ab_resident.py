#!/usr/bin/env python3
"""Same-box A/B of resident-kernel builds: alternates the given libraries / generations in child processes and prints
us per pivot of dense-LP(size,size,42) (devices differ by several per cent: never compare two boxes).
  python3 tools/ab_resident.py --size 2048 --rounds 3 name=lib.so[,GEN] ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("arms", nargs="+")
a = ap.parse_args()
res = {}
for r in range(a.rounds):
    for arm in a.arms:
        name, spec = arm.split("=")
        lib, _, gen = spec.partition(",")
        env = dict(os.environ, YALPS_HIP_LIB=os.path.join(ROOT, "yalps_amd", lib))
        if gen:
            env["YALPS_HIP_RESIDENT_GEN"] = gen
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_solve.py"), "--size", str(a.size), "--reps", "5"],
                             env=env, capture_output=True, text=True)
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        res.setdefault(name, []).append(round(rec["us_per_pivot"], 3))
        print(r, name, rec["kernel"], rec["pivots"], round(rec["us_per_pivot"], 3), flush=True)
print(json.dumps({"size": a.size, "us_per_pivot": res, "best": {k: min(v) for k, v in res.items()}}))
