#!/bin/bash
# Row shards, delayed row updates (dshard_kernel) against one sweep per pivot (wide_kernel in place), one GPU, same box:
#   tools/shard_measurements.sh OUTDIR      (run from the repo root on the GPU box)
set -x
out=${1:-gpurun_out/shard}; mkdir -p $out; export TMPDIR=/tmp
for dl in 1 0; do
  YALPS_HIP_SHARD_DELAY=$dl python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 2 --warmup 1 2> $out/shard16384_delay$dl.err | grep "^{" > $out/shard16384_delay$dl.json
  YALPS_HIP_SHARD_DELAY=$dl python3 bench.py --workload sharded --size 16384 --shard-rows 2048 --gpus 1 --steps 4 --warmup 1 2>/dev/null | grep "^{" > $out/shard2048x16384_delay$dl.json
  YALPS_HIP_SHARD_DELAY=$dl python3 bench.py --workload sharded --size 8192 --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep "^{" > $out/shard8192_delay$dl.json
  YALPS_HIP_SHARD_DELAY=$dl python3 bench.py --workload sharded --size 4096 --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep "^{" > $out/shard4096_delay$dl.json
done
for dp in 2 4 6 8; do # pivots per sweep, one rank of 8 (2048 rows) and one of 2 (8192 rows)
  YALPS_HIP_DELAY_DEPTH=$dp python3 bench.py --workload sharded --size 16384 --shard-rows 2048 --gpus 1 --steps 4 --warmup 1 2>/dev/null | grep "^{" > $out/depth${dp}_2048x16384.json
  YALPS_HIP_DELAY_DEPTH=$dp python3 bench.py --workload sharded --size 16384 --shard-rows 8192 --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep "^{" > $out/depth${dp}_8192x16384.json
done
rocprofv3 --kernel-trace --stats -d $out/shstats --output-format csv -- python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 1 --warmup 1 > /dev/null 2>&1
find $out/shstats -name "*kernel_stats.csv" -exec cp {} $out/shard16384_kernel_stats.csv \; ; rm -rf $out/shstats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/shpmc_$c --output-format csv -- python3 bench.py --workload sharded --size 16384 --gpus 1 --steps 1 --warmup 1 > /dev/null 2>&1
  python3 tools/pmc_summary.py $out/shpmc_$c $c dshard_kernel > $out/shard16384_$c.json; rm -rf $out/shpmc_$c
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload sharded --size 4096 --steps 1 --warmup 1 2> $out/rehearsal2.err | grep "^{" > $out/rehearsal2.json
echo finished
