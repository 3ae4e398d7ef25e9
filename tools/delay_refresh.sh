#!/bin/bash
# The delayed-update tables of DESIGN.md 4.9 again (same box, bounded launches) + BASELINE config 5 as a whole solve:
#   tools/delay_refresh.sh OUTDIR     (a part of tools/final_measurements.sh, for when only the delayed kernels changed)
set -x
out=${1:-gpurun_out/refresh}; mkdir -p $out; export TMPDIR=/tmp
: > $out/delay_table.txt
for shape in "--size 16384 --pivots 240" "--size 16384 --rows 4096 --pivots 480" "--size 16384 --rows 2048 --pivots 600" "--size 16384 --rows 1024 --pivots 1500" "--size 8192 --pivots 800" "--size 6000 --pivots 1200" "--size 5000 --pivots 2000" "--size 4096 --pivots 2000" "--size 1500 --rows 12000 --pivots 2000"; do
  for dl in 1 0; do YALPS_HIP_DELAY=$dl python3 tools/profile_solve.py $shape >> $out/delay_table.txt; done
done
python3 bench.py --size 16384 --steps 1 --warmup 0 --cpu-pivots 0 --sweep-launches 2 > $out/bench_16384.json 2> $out/bench_16384.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/pmc_$c --output-format csv -- python3 tools/profile_solve.py --size 16384 --pivots 240 > /dev/null 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c $c stream3_kernel > $out/delayed_16385_$c.json; rm -rf $out/pmc_$c
done
echo finished
