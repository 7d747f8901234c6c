#!/bin/bash
# round 3: k_read_sums with the qualities handed round by DPP: long-read tests, config 4 timing, kernel times
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
cd $R
python -m pytest tests -x -q -m gpu -k "long or nanopore or config4 or invalid or quality_bytes or getter" 2>&1 | tail -3 | tee $OUT/summary.txt
for v in "SQ_LONG_NW=8" "SQ_LONG=0"; do
  env $v python bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['checks'])"
done | tee -a $OUT/summary.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" | tee -a $OUT/summary.txt
rm -rf $OUT/stats
