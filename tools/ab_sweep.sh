#!/bin/bash
# same-box A/B of two library builds on the HBM-bound shapes: tools/ab_sweep.sh libA.so libB.so
cd "$(dirname "$0")/.."
for r in 1 2; do
  for lib in "$@"; do
    for shape in "--size 16384 --pivots 200" "--size 8192 --pivots 400" "--size 16384 --rows 1024 --pivots 1500" "--size 16384 --rows 4096 --pivots 400"; do
      YALPS_HIP_LIB=$PWD/yalps_amd/$lib python3 tools/profile_solve.py $shape | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['tableau'], d['kernel'], round(d['us_per_pivot'],2), round(d['algorithmic_TBps'],3))"
    done
  done
done
