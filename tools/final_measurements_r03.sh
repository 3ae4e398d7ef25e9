#!/bin/bash
# Round 3: everything the tables and profiles/r03_* are made from, in one GPU call (repo root, GPU box):
#   tools/final_measurements_r03.sh OUTDIR
set -x
out=${1:-gpurun_out/final3}; mkdir -p $out; export TMPDIR=/tmp
prof() { # prof <tag> <kernel substring for the PMC summary> -- <command ...>: rocprof kernel stats + separate FETCH_SIZE / WRITE_SIZE passes
  tag=$1; sub=$2; shift 3
  rocprofv3 --kernel-trace --stats -d $out/st_$tag --output-format csv -- "$@" > $out/${tag}.json 2> $out/${tag}.err
  find $out/st_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \; ; rm -rf $out/st_$tag
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/pmc_$tag --output-format csv -- "$@" > /dev/null 2>&1
    python3 tools/pmc_summary.py $out/pmc_$tag $c $sub > $out/${tag}_$c.json; rm -rf $out/pmc_$tag
  done
}
# headline (BASELINE config 2): the line, its rocprof summary
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py > $out/bench_prof.json 2> $out/bench_prof.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \; ; rm -rf $out/stats
echo "=== headline done"
# BASELINE config 5 on one GPU: whole solve, then bounded launches under the profiler (kernel stats + HBM traffic)
python3 bench.py --size 16384 --steps 1 --warmup 0 --cpu-pivots 0 --sweep-launches 2 > $out/bench_16384.json 2> $out/bench_16384.err
prof inplace_16385x16385 stream3_kernel -- python3 tools/profile_solve.py --size 16384 --pivots 320
prof inplace_8193x8193 stream3_kernel -- python3 tools/profile_solve.py --size 8192 --pivots 800
prof inplace_4097x16385 stream3_kernel -- python3 tools/profile_solve.py --size 16384 --rows 4096 --pivots 640
prof inplace_1025x16385 stream3_kernel -- python3 tools/profile_solve.py --size 16384 --rows 1024 --pivots 1500
echo "=== in place done"
python3 tools/shape_sweep.py 32x32 128x128 256x256 512x512 1024x1024 1536x1536 2048x2048 2560x2560 3072x3072 3300x3000 4096x4096 5000x5000 6000x6000 512x4096 4096x512 1000x6000 10000x1000 11000x900 12000x1500 1024x8000 256x8192 8192x8192 1024x16384 2048x16384 4096x16384 16384x16384 1000x20000 > $out/shape_sweep.txt 2>&1
echo "=== shapes done"
# row shards on one rank: the whole tableau and a rank's share of 2 / 4 / 8
for rows in 0 8192 4096 2048; do
  r=$( [ $rows = 0 ] && echo "" || echo "--shard-rows $rows" ); tag=$( [ $rows = 0 ] && echo 16385 || echo $(( rows + 1 )) )
  python3 bench.py --workload sharded --size 16384 $r --steps 3 --warmup 1 --pivots-per-step 256 2> /dev/null | grep "^{" > $out/shard_${tag}x16385.json
done
prof shard_2049x16385_prof dshard -- python3 bench.py --workload sharded --size 16384 --shard-rows 2048 --steps 2 --warmup 1 --pivots-per-step 256 --verify-pivots 0
prof shard_16385x16385_prof dshard -- python3 bench.py --workload sharded --size 16384 --steps 1 --warmup 1 --pivots-per-step 256 --verify-pivots 0
echo "=== shards done"
python3 tools/delayed_stages.py --kernel stream3 --size 16384 --pivots 480 --out $out/stream3_stages_16384.json > /dev/null 2>&1
python3 tools/delayed_stages.py --kernel stream3 --size 8192 --pivots 800 --out $out/stream3_stages_8192.json > /dev/null 2>&1
python3 tools/delayed_stages.py --kernel stream3 --size 4096 --pivots 800 --out $out/stream3_stages_4096.json > /dev/null 2>&1
python3 tools/delayed_stages.py --kernel dshard --size 16384 --rows 2048 --pivots 400 --out $out/dshard_stages_2049x16385.json > /dev/null 2>&1
python3 tools/delayed_stages.py --kernel dshard --size 16384 --pivots 320 --out $out/dshard_stages_16385x16385.json > /dev/null 2>&1
echo "=== stages done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 --sharded-c5-size 4096 2> $out/rehearsal_2ranks.err | grep "^{" > $out/rehearsal_2ranks.json
python3 tools/netlib_paths.py > $out/netlib_paths.txt 2>&1
echo finished
