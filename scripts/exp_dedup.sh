#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/dd
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dd -- python3 $R/scripts/exp_dedup.py 2>&1 | grep " ms"
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob("$R/gpurun_out/dd/**/*kernel_stats.csv",recursive=True),key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
