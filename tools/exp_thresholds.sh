#!/bin/bash
# experiment: panel sweep against the sweep straight from L2 for workgroups with few rows, by depth (stream3: whole solves; dshard: one rank)
for shape in 1024x16384 2048x16384 1024x8000 3000x16384; do
  for panel in 0 1; do
    for depth in 8 12 16; do
      echo "stream3 $shape panel=$panel depth=$depth: $(YALPS_HIP_STREAM3_PANEL=$panel YALPS_HIP_DELAY_DEPTH=$depth python3 tools/shape_sweep.py $shape 2>/dev/null | grep "^{'" | cut -c1-200)"
    done
  done
done
for rows in 2048 4096 6000; do
  for panel in 0 1; do
    for depth in 8 12 16; do
      echo "dshard rows=$rows panel=$panel depth=$depth: $(YALPS_HIP_SHARD_PANEL=$panel YALPS_HIP_DELAY_DEPTH=$depth python3 bench.py --workload sharded --size 16384 --shard-rows $rows --steps 3 --warmup 1 --pivots-per-step 256 --verify-pivots 0 2>/dev/null | grep "^{" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['roofline']['us_per_pivot'],2), r['roofline']['kernel'])")"
    done
  done
done
