#!/bin/bash
# round 3: are the length-sorted k_span launches bound by latency or by bandwidth?  kernel times per window count
# with 8 / 12 / 16 waves per workgroup (SQ_SPAN_WAVES)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3n
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for w in 8 12 16; do
  SQ_SPAN_WAVES=$w rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$w -- python3 $R/scripts/bench_ragged.py 25000000 50 > $OUT/run$w.txt 2>&1
  f=$(find $OUT/st$w -name "*kernel_stats.csv" | head -1)
  echo "== SQ_SPAN_WAVES=$w: $(grep lengths $OUT/run$w.txt)"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:9]:
    if 'k_span' in r['Name'] or 'rocprim' in r['Name']: print('  ', r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')"
  rm -rf $OUT/st$w
done | tee $OUT/summary.txt
