#!/bin/bash
# One-rank delayed-shard solves against the oracle (tools/dbg_shard.py), forced and natural shapes.  usage: tools/dbg_dshard.sh [outfile]
out=${1:-gpurun_out/dshard/dbg.txt}
mkdir -p "$(dirname "$out")"
run() { # env... -- M N seed pivots
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    echo "== ${envs[*]} $*" >> "$out"
    env "${envs[@]}" timeout -k 10 240 python tools/dbg_shard.py "$@" >> "$out" 2>&1 || echo "FAILED rc=$?" >> "$out"
}
: > "$out"
run YALPS_HIP_DELAY_MIN_ROWS=1 -- 300 3000 2 8
run YALPS_HIP_DELAY_MIN_ROWS=1 -- 300 3000 3 50
run YALPS_HIP_DELAY_MIN_ROWS=1 YALPS_HIP_DELAY_DEPTH=3 -- 300 6000 5 77
run YALPS_HIP_DELAY_MIN_ROWS=1 -- 300 9000 3 57
run YALPS_HIP_DELAY_MIN_ROWS=1 YALPS_HIP_DELAY_DEPTH=8 -- 700 16384 4 30
run A=1 -- 1500 5000 5 100
run A=1 -- 4000 4500 7 200
run YALPS_HIP_SHARD_DELAY=0 -- 1500 5000 5 100
cat "$out"
