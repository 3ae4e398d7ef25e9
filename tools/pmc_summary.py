#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc pass (counter_collection csv under a directory) per kernel name:
  python3 tools/pmc_summary.py <dir> <COUNTER> [kernel substring]  ->  JSON {kernel: {dispatches, sum, avg}}"""
import csv
import glob
import json
import os
import sys

d, counter = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
out = {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter or sub not in row.get("Kernel_Name", ""):
                continue
            k = row["Kernel_Name"]
            e = out.setdefault(k, {"dispatch_ids": set(), "sum": 0.0})
            e["dispatch_ids"].add(row.get("Dispatch_Id"))
            e["sum"] += float(row["Counter_Value"])
print(json.dumps({k: {"dispatches": len(v["dispatch_ids"]), "sum": v["sum"], "avg_per_dispatch": v["sum"] / max(len(v["dispatch_ids"]), 1)}
                  for k, v in out.items()}, indent=1))
