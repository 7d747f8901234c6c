#!/bin/bash
# round 3: the rows of a ragged batch by the batch's length counts (k_span_scatter) instead of a radix sort
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3q
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_staging.py tests/test_parser_golden.py tests/test_bam_golden.py -q -m gpu -x -k "sorted or length_counts or ragged or staging or trimmed or split or bam" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | tee $OUT/summary.txt; grep -B30 "^E " $OUT/tests.log | head -80 | tee -a $OUT/summary.txt
for v in "SQ_X=0" "SQ_SPAN_RADIX=1" "SQ_X=0" "SQ_SPAN_RADIX=1"; do
  env $v python scripts/bench_ragged.py 25000000 50 | tail -1 | sed "s/^/$v /"
done | tee -a $OUT/summary.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 $R/scripts/bench_ragged.py 25000000 50 > /dev/null 2>&1
f=$(find $OUT/st -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:10]:
    print('  ', r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" | tee -a $OUT/summary.txt
rm -rf $OUT/st
