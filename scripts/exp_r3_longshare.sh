#!/bin/bash
# round 3: k_span<LONG>: workgroup shares of equal cost (a new segment costs SQ_LONG_STRETCH_COST spans; 0: equal numbers of spans)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3w
mkdir -p $OUT
cd $R
for v in 0 6 10 12 16 20 32 48 12 16; do
  SQ_LONG_STRETCH_COST=$v python bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stretch cost $v', d['value'], d['ms_per_step'], all(d['checks'].values()))"
done | tee -a $OUT/summary.txt
cd /tmp; export TMPDIR=/tmp
SQ_LONG_STRETCH_COST=16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 $R/bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
f=$(find $OUT/st -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:6]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" | tee -a $OUT/summary.txt
rm -rf $OUT/st
