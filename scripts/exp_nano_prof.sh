#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/nano_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $R/bench.py --kind nanopore --steps 2 --warmup 1 --cpu-sample 0 > $OUT/log 2>&1
grep -o '"value": [0-9.]*' $OUT/log | head -1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/s/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['AverageNs'])/1e6,3), r['Percentage'])
PY
