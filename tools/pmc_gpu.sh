#!/bin/bash
# One rocprofv3 --pmc pass per counter group over a short bench run (GPU box).
#   tools/pmc_gpu.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'anonymous namespace' not in k: continue
        k=k[k.index('namespace)::')+12:].split('(')[0][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-28s %14.1f  (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
