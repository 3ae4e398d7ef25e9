export TMPDIR=/tmp
for rows in 0 2048; do
  r=$( [ $rows = 0 ] && echo "" || echo "--shard-rows $rows" )
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sh$rows --output-format csv -- python3 bench.py --workload sharded --size 16384 $r --steps 2 --warmup 1 --pivots-per-step 256 --verify-pivots 0 > /dev/null 2>&1
  find gpurun_out/prof_sh$rows -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_sh${rows}_stats.csv \; ; rm -rf gpurun_out/prof_sh$rows
done
