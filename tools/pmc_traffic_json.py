#!/usr/bin/env python3
"""HBM traffic per pivot of the delayed-update kernels from the separate rocprofv3 --pmc passes of tools/final_measurements_r03.sh:
  python3 tools/pmc_traffic_json.py <final dir> > profiles/r03_pmc_traffic_delayed.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming reads
(MI355X_MICROARCH.md, HBM / rocprofv3 section) and is doubled; WRITE_SIZE is exact.  Per launch = per dispatch of the kernel;
per pivot = / the pivots of the bounded launch (profile_solve.py) resp. of all step launches (bench.py --workload sharded)."""
import json
import os
import sys

d = sys.argv[1]
HBM_PEAK = 8e12


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def pmc(tag, counter, sub):
    rec = json.load(open(os.path.join(d, "%s_%s.json" % (tag, counter))))
    return sum(v["sum"] for k, v in rec.items() if sub in k), sum(v["dispatches"] for k, v in rec.items() if sub in k)


cases = []
for tag in ("inplace_16385x16385", "inplace_8193x8193", "inplace_4097x16385", "inplace_1025x16385"):
    if not os.path.exists(os.path.join(d, tag + ".json")):
        continue
    run = last_json(os.path.join(d, tag + ".json"))
    fetch, _ = pmc(tag, "FETCH_SIZE", "stream3_kernel")
    write, _ = pmc(tag, "WRITE_SIZE", "stream3_kernel")
    piv = run["pivots"]
    rd, wr = 2.0 * fetch * 1024 / piv, write * 1024 / piv
    cases.append({"tableau": run["tableau"], "kernel": run["kernel"], "pivots_per_launch": piv, "us_per_pivot_hip_events": run["us_per_pivot"],
                  "algorithmic_bytes_per_pivot": run["algorithmic_bytes_per_pivot"], "algorithmic_equiv_TBps": run["algorithmic_TBps"],
                  "read_bytes_per_pivot": rd, "write_bytes_per_pivot": wr, "traffic_bytes_per_pivot": rd + wr,
                  "traffic_over_algorithmic": (rd + wr) / run["algorithmic_bytes_per_pivot"],
                  "hbm_TBps": (rd + wr) / (run["us_per_pivot"] * 1e-6) / 1e12, "hbm_frac_of_8TBps": (rd + wr) / (run["us_per_pivot"] * 1e-6) / HBM_PEAK})
for tag, rows in (("shard_2049x16385_prof", 2049), ("shard_16385x16385_prof", 16385)):
    if not os.path.exists(os.path.join(d, tag + ".json")):
        continue
    run = last_json(os.path.join(d, tag + ".json"))
    rd = wr = 0.0
    launches = 0
    for sub in ("dshard_kernel", "dshard_select_kernel", "dshard_sweep_kernel"):  # (the sweep kernel: one launch per `depth` pivots, its bytes spread over the pivots)
        f, n = pmc(tag, "FETCH_SIZE", sub + "<")
        w, _ = pmc(tag, "WRITE_SIZE", sub + "<")
        rd += 2.0 * f * 1024
        wr += w * 1024
        if sub != "dshard_sweep_kernel":
            launches = max(launches, n)
    bpp = 16 * (rows - 1) * 16385 + 16 * 16385 + 8 * (rows - 1) + 8 * 16384 + 16 * (rows - 1)
    cases.append({"tableau": "%dx16385 (row shard, one rank)" % rows, "kernel": run["roofline"]["kernel"], "step_launches_profiled": launches,
                  "us_per_pivot_wall": run["roofline"]["us_per_pivot"], "algorithmic_bytes_per_pivot": bpp,
                  "read_bytes_per_pivot": rd / launches, "write_bytes_per_pivot": wr / launches, "traffic_bytes_per_pivot": (rd + wr) / launches,
                  "traffic_over_algorithmic": (rd + wr) / launches / bpp,
                  "hbm_TBps": (rd + wr) / launches / (run["roofline"]["us_per_pivot"] * 1e-6) / 1e12,
                  "note": "select + step (+ sweep) kernels; launches counted by the profiler include the warm-up pivots, the per-launch traffic does not depend on that"})
print(json.dumps({"what": "delayed-update kernels, round 3 (panel sweep, up to 16 pending pivots): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                          "(tools/final_measurements_r03.sh); KB units; FETCH_SIZE doubled (gfx950, 16 B/lane reads); per pivot", "cases": cases}, indent=1))
