#!/bin/bash
# round 3: config 4 through k_span<LONG>: 4 and 8 windows per segment, k_seg as the baseline; kernel times
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3i
mkdir -p $OUT
cd $R
for v in "SQ_LONG_NW=8" "SQ_LONG_NW=4" "SQ_LONG=0"; do
  env $v python bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['checks'])"
done | tee $OUT/summary.txt
cd /tmp; export TMPDIR=/tmp
for v in 8 4; do
  SQ_LONG_NW=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$v -- python3 $R/bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
  f=$(find $OUT/stats$v -name "*kernel_stats.csv" | head -1)
  echo "== SQ_LONG_NW=$v" | tee -a $OUT/summary.txt
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" | tee -a $OUT/summary.txt
  rm -rf $OUT/stats$v
done
