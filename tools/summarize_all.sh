#!/bin/bash
# After tools/profile_gpu.sh r03_<workload> runs on the GPU box: profiles/r03_<workload>.md and the entries of
# profiles/traffic.json for the workloads whose directories are under gpurun_out/.
#   tools/summarize_all.sh [tag prefix, default r03]
cd "$(dirname "$0")/.."
P=${1:-r03}
for spec in "cfg2 25x10" "cfg3 25x10" "cfg4 25x10" "cfg2h3 25x10x5" "cfg5 25x10x5"; do
  set -- $spec
  [ -d gpurun_out/prof_${P}_$1 ] && python tools/summarize_profile.py gpurun_out/prof_${P}_$1 ${P}_$1 $1 4096 $2 > /dev/null && echo "$1 summarized"
done
