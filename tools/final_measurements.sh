#!/bin/bash
# Everything the round's tables and profiles/ are made from, in one GPU call (run from the repo root on the GPU box):
#   tools/final_measurements.sh OUTDIR
set -x
out=${1:-gpurun_out/final}; mkdir -p $out; export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py > $out/bench_prof.json 2> $out/bench_prof.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \; ; rm -rf $out/stats
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py --steps 2 --warmup 0 --cpu-pivots 0 --sweep-launches 4 > /dev/null 2> $out/pmc_fetch.err
python3 tools/pmc_summary.py $out/pmc_fetch FETCH_SIZE > $out/fetch_summary.json; rm -rf $out/pmc_fetch
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py --steps 2 --warmup 0 --cpu-pivots 0 --sweep-launches 4 > /dev/null 2> $out/pmc_write.err
python3 tools/pmc_summary.py $out/pmc_write WRITE_SIZE > $out/write_summary.json; rm -rf $out/pmc_write
python3 tools/resident_stages.py --size 2048 > $out/stages_2048.json 2> $out/stages.err
YALPS_HIP_RESIDENT_GEN=1 python3 tools/resident_stages.py --size 2048 > $out/stages_2048_gen1.json 2>> $out/stages.err
python3 tools/shape_sweep.py 32x32 128x128 256x256 512x512 1024x1024 1536x1536 2048x2048 2560x2560 3072x3072 3300x3000 4096x4096 5000x5000 512x4096 4096x512 1000x6000 10000x1000 11000x900 12000x1500 1024x8000 256x8192 8192x8192 1024x16384 1000x20000 > $out/shape_sweep.txt 2>&1
# delayed row updates (stream2_kernel / stream3_kernel) against one sweep per pivot, same box, bounded launches
: > $out/delay_table.txt
for shape in "--size 16384 --pivots 240" "--size 16384 --rows 4096 --pivots 480" "--size 16384 --rows 2048 --pivots 600" "--size 16384 --rows 1024 --pivots 1500" "--size 8192 --pivots 800" "--size 6000 --pivots 1200" "--size 5000 --pivots 2000" "--size 4096 --pivots 2000" "--size 1500 --rows 12000 --pivots 2000"; do
  for dl in 1 0; do YALPS_HIP_DELAY=$dl python3 tools/profile_solve.py $shape >> $out/delay_table.txt; done
done
python3 bench.py --size 16384 --steps 1 --warmup 0 --cpu-pivots 0 --sweep-launches 2 > $out/bench_16384.json 2> $out/bench_16384.err
python3 tools/netlib_paths.py > $out/netlib_paths.txt 2>&1
YALPS_HIP_DELAY=0 python3 tools/netlib_paths.py > $out/netlib_paths_nodelay.txt 2>&1
python3 bench_bnb.py > $out/bnb.json 2> $out/bnb.err
python3 bench_table.py > $out/table.json 2> $out/table.err
# row shards: one sweep per pivot (wide_kernel in place; what profiles/r02_sharded_16384* hold) ...
YALPS_HIP_SHARD_DELAY=0 python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 2 --warmup 1 2> $out/shard16384.err | grep "^{" > $out/shard16384.json
YALPS_HIP_SHARD_DELAY=0 python3 bench.py --workload sharded --size 4096 --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep "^{" > $out/shard4096.json
YALPS_HIP_SHARD_DELAY=0 rocprofv3 --kernel-trace --stats -d $out/shstats --output-format csv -- python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 1 --warmup 1 > /dev/null 2>&1
find $out/shstats -name "*kernel_stats.csv" -exec cp {} $out/shard16384_kernel_stats.csv \; ; rm -rf $out/shstats
for c in FETCH_SIZE WRITE_SIZE; do
  YALPS_HIP_SHARD_DELAY=0 rocprofv3 --pmc $c -d $out/shpmc_$c --output-format csv -- python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 1 --warmup 1 > /dev/null 2>&1
  python3 tools/pmc_summary.py $out/shpmc_$c $c wide_kernel > $out/shard16384_$c.json; rm -rf $out/shpmc_$c
done
# ... and with delayed row updates (dshard_kernel, the default): profiles/r02_shard_delay_*
tools/shard_measurements.sh $out/shard_delay
# sweep_kernel on the HBM-bound shapes: rocprof kernel stats + PMC passes of one bounded launch each
for shape in "16384 0 300" "8192 0 800" "16384 1024 2000" "16384 4096 600"; do
  set -- $shape; tag=$( [ "$2" = 0 ] && echo $(( $1 + 1 ))x$(( $1 + 1 )) || echo $(( $2 + 1 ))x$(( $1 + 1 )) )
  rows=$( [ "$2" = 0 ] || echo "--rows $2" )
  rocprofv3 --kernel-trace --stats -d $out/sw_$tag --output-format csv -- python3 tools/profile_solve.py --size $1 $rows --pivots $3 > $out/sweep_$tag.json 2> /dev/null
  find $out/sw_$tag -name "*kernel_stats.csv" -exec cp {} $out/sweep_${tag}_kernel_stats.csv \; ; rm -rf $out/sw_$tag
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/swpmc --output-format csv -- python3 tools/profile_solve.py --size $1 $rows --pivots $3 > /dev/null 2>&1
    python3 tools/pmc_summary.py $out/swpmc $c _kernel > $out/sweep_${tag}_$c.json; rm -rf $out/swpmc
  done
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 2> $out/rehearsal_replicas2.err | grep "^{" > $out/rehearsal_replicas2.json
python3 tools/node_overhead.py "Monster 2" > $out/node_overhead.txt 2>&1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload sharded --size 4096 --steps 1 --warmup 1 2> $out/rehearsal2.err | grep "^{" > $out/rehearsal2.json
echo finished
