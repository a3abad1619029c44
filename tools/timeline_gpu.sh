#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of a short bench run, then the launches of the last
# training step with their start offsets and durations -> gpurun_out/tl/timeline.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 20 "$@" > $OUT/trace.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the launch that ends a step (the update: a kernel of its own, or the reduction of the one-kernel backward pass);
# the last one = end of the last step; walk back to the one before
last='bwd_reduce_kernel' if any('bwd_reduce_kernel' in r['Kernel_Name'] for r in rows) and not any('adam_kernel' in r['Kernel_Name'] for r in rows[-40:]) else 'adam_kernel'
idx=[i for i,r in enumerate(rows) if last in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['End_Timestamp'])
out=open('$OUT/timeline.txt','w')
for r in rows[a+1:b+1]:
    n=r['Kernel_Name']
    n=n[n.find('::')+2:] if '::' in n else n
    out.write('%8.1f %8.1f  q%s %s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),n[:70]))
out.close()
PY
