# kernel trace of a workload's training step: tools/trace_train.sh <tag> [bench args]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_EAGER_TRAIN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 10 "$@" > $OUT/log.txt 2>&1
