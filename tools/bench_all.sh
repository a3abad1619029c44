#!/bin/bash
# Runs on the GPU box: the benchmark line of every workload -> gpurun_out/bench_<workload>.json
#   tools/bench_all.sh [extra bench.py arguments]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for w in cfg2 cfg3 cfg4 cfg5 cfg2h3 cfg2w64 ml1m; do
  timeout -k 10 400 python3 $ROOT/bench.py --workload $w --no-cpu-baseline "$@" > $ROOT/gpurun_out/bench_$w.json 2> $ROOT/gpurun_out/bench_$w.err || { echo "$w failed"; tail -5 $ROOT/gpurun_out/bench_$w.err; exit 1; }
  echo "$w done"
done
