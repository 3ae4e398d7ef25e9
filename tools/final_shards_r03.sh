set -x
out=gpurun_out/final6; mkdir -p $out; export TMPDIR=/tmp
prof() { tag=$1; sub=$2; shift 3
  rocprofv3 --kernel-trace --stats -d $out/st_$tag --output-format csv -- "$@" > $out/${tag}.json 2> $out/${tag}.err
  find $out/st_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \; ; rm -rf $out/st_$tag
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/pmc_$tag --output-format csv -- "$@" > /dev/null 2>&1
    python3 tools/pmc_summary.py $out/pmc_$tag $c $sub > $out/${tag}_$c.json; rm -rf $out/pmc_$tag
  done
}
for rows in 0 8192 4096 2048; do
  r=$( [ $rows = 0 ] && echo "" || echo "--shard-rows $rows" ); tag=$( [ $rows = 0 ] && echo 16385 || echo $(( rows + 1 )) )
  python3 bench.py --workload sharded --size 16384 $r --steps 3 --warmup 1 --pivots-per-step 256 2> /dev/null | grep "^{" > $out/shard_${tag}x16385.json
done
prof shard_2049x16385_prof dshard -- python3 bench.py --workload sharded --size 16384 --shard-rows 2048 --steps 2 --warmup 1 --pivots-per-step 256 --verify-pivots 0
prof shard_16385x16385_prof dshard -- python3 bench.py --workload sharded --size 16384 --steps 1 --warmup 1 --pivots-per-step 256 --verify-pivots 0
python3 tools/delayed_stages.py --kernel dshard --size 16384 --rows 2048 --pivots 400 --out $out/dshard_stages_2049x16385.json > /dev/null 2>&1
python3 tools/delayed_stages.py --kernel dshard --size 16384 --pivots 320 --out $out/dshard_stages_16385x16385.json > /dev/null 2>&1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 --sharded-c5-size 4096 2> $out/rehearsal_2ranks.err | grep '^{' > $out/rehearsal_2ranks.json
echo finished
