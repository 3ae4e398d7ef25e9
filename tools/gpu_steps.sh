#!/bin/bash
# Runs GPU steps one after the other on the GPU box, each under its own `timeout -k 10`; a step that fails with an
# ordinary error (an assertion, a non-zero exit) does not stop the ones after it, a step that was TIMED OUT or KILLED does:
# nothing else is started on a GPU that may be hung.
#   usage: tools/gpu_steps.sh "<seconds> <logfile> <command ...>" ...
set -u
mkdir -p gpurun_out
overall=0
for step in "$@"; do
    secs=${step%% *}; rest=${step#* }
    log=${rest%% *}; cmd=${rest#* }
    echo "=== [$secs s] $cmd > $log"
    ( while sleep 60; do echo "    ... $(date +%T) still running: $(tail -c 120 "$log" 2>/dev/null | tr '\n' ' ')"; done ) &
    beat=$!
    timeout -k 10 "$secs" bash -c "$cmd" > "$log" 2>&1
    rc=$?
    kill $beat 2>/dev/null; wait $beat 2>/dev/null
    echo "=== rc=$rc"
    tail -n 6 "$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== step timed out or was killed: stopping here"
        exit $rc
    fi
    [ $rc -ne 0 ] && overall=$rc
done
exit $overall
