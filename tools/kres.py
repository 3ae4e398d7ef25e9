#!/usr/bin/env python3
"""Register / scratch use of the kernels in a built object or library (the gfx950 code object's metadata).
usage: python tools/kres.py <file.o|.so> [substring]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalps_amd import build  # noqa: E402

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
ks = build.kernel_metadata(path)
for name in sorted(ks):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if sub and sub not in dem:
        continue
    md = ks[name]
    print("%-44s vgpr %3s agpr %2s sgpr %3s vgpr_spill %3s sgpr_spill %3s scratch %4s lds %6s" % (
        dem, md.get("vgpr_count"), md.get("agpr_count"), md.get("sgpr_count"), md.get("vgpr_spill_count"), md.get("sgpr_spill_count"),
        md.get("private_segment_fixed_size"), md.get("group_segment_fixed_size")))
