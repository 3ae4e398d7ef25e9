#!/bin/bash
# experiment: a shard's sweep inside the step kernel (0) / as a launch of its own (1), one rank, 16385 columns
for rows in 0 8192 4096 2048 1024; do
  for x in 0 1; do
    r=$( [ $rows = 0 ] && echo "" || echo "--shard-rows $rows" )
    echo "rows=$rows xsweep=$x: $(YALPS_HIP_SHARD_XSWEEP=$x python3 bench.py --workload sharded --size 16384 $r --steps 3 --warmup 1 --pivots-per-step 256 --verify-pivots 0 2>/dev/null | grep "^{" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['roofline']['us_per_pivot'],2), r['roofline']['kernel'])")"
  done
done
