#!/bin/bash
# round 3: config 3's kernels by batch size (fixed costs per launch?)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3y
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in 25000000 12500000 6250000 3125000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$n -- python3 $R/scripts/bench_config3.py $n 3 > $OUT/run$n.txt 2>&1
  f=$(find $OUT/st$n -name "*kernel_stats.csv" | head -1)
  echo "== $n pairs: $(tail -1 $OUT/run$n.txt)"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:9]:
    if 'synth' not in r['Name']: print('  ', r['Name'][:60].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')"
  rm -rf $OUT/st$n
done | tee $OUT/summary.txt
