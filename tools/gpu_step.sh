#!/bin/bash
# Runs GPU steps one after the other; stops at the first step that ends in anything
# other than success or an ordinary test failure (a fault, an abort, a time limit).
#   tools/gpu_step.sh "<seconds> <log> <command...>" ...
mkdir -p gpurun_out
for step in "$@"; do
    set -- $step
    limit=$1; log=$2; shift 2
    echo "== $* (limit ${limit}s) -> $log"
    timeout -k 10 "$limit" "$@" > "$log" 2>&1
    rc=$?
    echo "== rc $rc"
    tail -n 25 "$log"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
        echo "== stopping: step ended with status $rc"
        exit $rc
    fi
done
exit 0
