# Runs on the GPU box: the bench line of the library in the tree and of every variant build named in $VARIANTS
# (graphsage_amd/variants/libgsage_<name>.so, see tools/build_variant.sh), with $BENCH_ARGS.
#   gpurun -- 'VARIANTS="a b" BENCH_ARGS="--workload cfg3" bash tools/run_v.sh > gpurun_out/v.log 2>&1'
set -e
cd $GRAFT_REPO_ROOT
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export GSAGE_LIB=$GRAFT_REPO_ROOT/graphsage_amd/variants/libgsage_$v.so; fi
  echo "== variant ${v:-default}"
  timeout -k 10 200 python bench.py --no-cpu-baseline $BENCH_ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step ms', d['ms_per_step'], 'kernel us', d['roofline'].get('launch_us'), 'train ms', d.get('train_step',{}).get('ms_per_step'))"
done
