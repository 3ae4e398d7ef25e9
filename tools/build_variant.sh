#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  ->  yalps_amd/libyalps_hip_NAME.so: the resident2 translation units rebuilt with the
# extra flags, everything else from yalps_amd/build (run `python -c "from yalps_amd import build; build.build_hip()"` first).
# For same-box A/B measurements (tools/ab_resident.py); the variants are not shipped.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p /tmp/yalps_var_$name
for u in a b; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c -o /tmp/yalps_var_$name/r2$u.o yalps_amd/csrc/persistent_resident2_$u.hip &
done
wait
ls yalps_amd/build/*.o | grep -v persistent_resident2_ | xargs /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o yalps_amd/libyalps_hip_$name.so /tmp/yalps_var_$name/r2a.o /tmp/yalps_var_$name/r2b.o
echo built yalps_amd/libyalps_hip_$name.so
