#!/bin/bash
# Shader clock under two loads (GPU box): GRBM_GUI_ACTIVE per kernel against its duration.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -- $ROOT/tools/micro/mfma_peak 1 > $OUT/mfma.log 2>&1 || { echo mfma failed; tail -3 $OUT/mfma.log; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/wide -- python3 $ROOT/tools/time_wide.py > $OUT/wide.log 2>&1 || { echo wide failed; tail -3 $OUT/wide.log; }
python3 - <<PY
import csv,glob
for tag,pat in (('mfma','mfma_loop'),('wide','sage_wide_kernel')):
    dur={}
    for f in glob.glob('$OUT/%s/*/*kernel_trace.csv'%tag):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']: dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    cyc={}
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv'%tag):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE': cyc[r['Dispatch_Id']]=float(r['Counter_Value'])
    for d in sorted(cyc, key=int)[-3:]:
        if d in dur: print(tag, 'dispatch', d, 'ns', dur[d], 'GRBM_GUI_ACTIVE', cyc[d], 'per ns /8 =', cyc[d]/8/dur[d])
PY
