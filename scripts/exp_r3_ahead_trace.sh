#!/bin/bash
# memory-copy and kernel trace of the pinned device-split path with the upload sent ahead
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/exp_e2e_timeline.py > $OUT/trace_run.txt 2>&1
python3 - <<PY
import csv,glob
mc=glob.glob('$OUT/trace/**/*memory_copy_trace.csv',recursive=True)[0]
kt=glob.glob('$OUT/trace/**/*kernel_trace.csv',recursive=True)[0]
ev=[]
for r in csv.DictReader(open(mc)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r['Direction']))
for r in csv.DictReader(open(kt)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40]))
ev.sort()
t0=ev[0][0]
big=[e for e in ev if e[1]-e[0]>200000 or 'COPY' in e[2] and e[1]-e[0]>100000]
with open('$OUT/trace_summary.txt','w') as f:
    for s,e,n in ev:
        if e-s>50000: f.write(f"{(s-t0)/1e6:10.3f} ms  +{(e-s)/1e6:7.3f} ms  {n}\n")
PY
rm -rf $OUT/trace
tail -5 $OUT/trace_run.txt
