#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats plus two PMC passes
# (FETCH_SIZE, WRITE_SIZE need separate runs) of the default bench workload.
#   tools/profile_gpu.sh <tag> [extra bench.py args]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one launch at a time in every pass: with two batches in flight a launch's own duration says nothing about the kernel
export BENCH_NO_PIPELINE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/write.log 2>&1 || exit 3
# optional fourth pass (GS_PROFILE_SQ=1): wave-state and MFMA counters, with the GPU-active cycle count
if [ "${GS_PROFILE_SQ:-0}" = "1" ]; then
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1 || exit 4
fi
grep '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
ls $OUT/*/*
