#!/bin/bash
# round 3: do the length-sorted launches have a fixed cost?  kernel times per window count for 25 M / 12.5 M / 6.25 M / 3.1 M reads
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3u
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in 25000000 12500000 6250000 3125000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$n -- python3 $R/scripts/bench_ragged.py $n 50 > $OUT/run$n.txt 2>&1
  f=$(find $OUT/st$n -name "*kernel_stats.csv" | head -1)
  echo "== $n reads: $(grep lengths $OUT/run$n.txt)"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: r['Name']):
    if 'k_span' in r['Name']: print('  ', r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg', round(float(r['MinNs'])/1e6,3), 'min')"
  rm -rf $OUT/st$n
done | tee $OUT/summary.txt
