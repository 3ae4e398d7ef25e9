#!/bin/bash
# experiment: what a sweep of stream3_kernel costs as a function of the pending pivots (is there a constant per sweep?)
out=${1:-gpurun_out/sweepconst}; mkdir -p $out
for size in 4096 8192; do
  for panel in 1 0; do
    for depth in 4 8 12 16; do
      YALPS_HIP_STREAM3_PANEL=$panel YALPS_HIP_DELAY_DEPTH=$depth python3 tools/delayed_stages.py --kernel stream3 --size $size --pivots 480 --out $out/s${size}_p${panel}_d${depth}.json > /dev/null 2>&1
      python3 - $out/s${size}_p${panel}_d${depth}.json $size $panel $depth <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); st={s['id']:s['us_mean'] for s in r['stages']}
d=int(sys.argv[4]); sw=st.get(10,0)+st.get(8,0)
print("size %s panel %s depth %2d kernel %-28s us/pivot %.2f  stage8 %.2f stage10 %.2f  (8+10)*depth = %.1f us per sweep" % (sys.argv[2],sys.argv[3],d,r['kernel'],r['stamped_us_per_pivot'],st.get(8,0),st.get(10,0),sw*d))
PY
    done
  done
done
