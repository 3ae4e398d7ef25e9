#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  ->  yalps_amd/libyalps_hip_NAME.so: the named persistent_*.hip translation units (default: resident2_a resident2_b) rebuilt with the
# extra flags, everything else from yalps_amd/build (run `python -c "from yalps_amd import build; build.build_hip()"` first).
# For same-box A/B measurements (tools/ab_resident.py); the variants are not shipped.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p /tmp/yalps_var_$name
units=${3:-"resident2_a resident2_b"}   # third argument: which persistent_*.hip units to rebuild (default: generation 2)
objs=""; skip=""
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c -o /tmp/yalps_var_$name/$u.o yalps_amd/csrc/persistent_$u.hip &
  objs="$objs /tmp/yalps_var_$name/$u.o"; skip="$skip -e persistent_$u\\."
done
wait
ls yalps_amd/build/*.o | grep -v $skip | xargs /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o yalps_amd/libyalps_hip_$name.so $objs
echo built yalps_amd/libyalps_hip_$name.so
