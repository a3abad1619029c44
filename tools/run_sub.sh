# Runs on the GPU box: the three-hop route's parity tests, then its bench line.
#   gpurun -- 'bash tools/run_sub.sh > gpurun_out/sub.log 2>&1'
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k subtree 2>&1 | tail -15
timeout -k 10 200 python bench.py --workload cfg2h3 --no-cpu-baseline --train-steps 0
