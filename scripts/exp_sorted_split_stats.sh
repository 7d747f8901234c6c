#!/bin/bash
# round 5: the length-sorted route per window count, one wave for both streams against a wave per stream -- the kernels' own
# durations (rocprofv3 --kernel-trace --stats over bench.py --configs ragged_50_150 / a 100..250-base variant is not in bench: 50..150 only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/exp_sorted_split
mkdir -p $OUT
: > $OUT/stats.txt
for s in 0 1; do
  SQ_SPAN_SORTED_SPLIT=$s rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$s -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --configs ragged_50_150 > $OUT/st$s.json 2> $OUT/st$s.err
  echo "== SQ_SPAN_SORTED_SPLIT=$s: kernel, calls, average ns" >> $OUT/stats.txt
  find $OUT/st$s -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    if 'k_span' in r['Name'] and 'true, true' in r['Name'] or 'scatter' in r['Name']:
        print(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60], r['Calls'], round(float(r['AverageNs'])))
" >> $OUT/stats.txt
  rm -rf $OUT/st$s
done
cat $OUT/stats.txt
