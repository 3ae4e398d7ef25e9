#!/bin/bash
# experiment: pending pivots of the 16-unit form (panels of 768 columns): whole solves by depth
for shape in 16384x16384 4096x16384 3000x16384; do
  for depth in 16 20 22; do
    echo "depth=$depth: $(YALPS_HIP_DELAY_DEPTH=$depth python3 tools/shape_sweep.py $shape 2>/dev/null | grep "^{'" | cut -c1-200)"
  done
done
