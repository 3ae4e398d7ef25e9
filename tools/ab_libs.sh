#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh "<shapes>" libA.so libB.so  (alternating, twice)
shapes=$1; shift
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"; YALPS_HIP_LIB=$PWD/yalps_amd/$lib python3 tools/shape_sweep.py $shapes 2>/dev/null | grep "^{'" | cut -c1-170
done; done
