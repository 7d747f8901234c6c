#!/bin/bash
# round 3: the length-sorted launches with the DMA alone (nothing counted: SQ_SPAN_PROBE=2) and with the counting alone
# (no DMA issued, stale slots: SQ_SPAN_PROBE=1); probe library scripts/build/libsqgpu_probe.so; one wave for both streams
# (the stamped build of a wave per stream spills)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3t
mkdir -p $OUT
cd $R
cp scripts/build/libsqgpu_probe.so sequali_amd/libsqgpu.so
cd /tmp; export TMPDIR=/tmp
for v in "-1" "2" "1"; do
  SQ_SPAN_SPLIT=0 SQ_SPAN_PROBE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$v -- python3 $R/scripts/bench_ragged.py 25000000 50 > $OUT/run$v.txt 2>&1
  f=$(find $OUT/st$v -name "*kernel_stats.csv" | head -1)
  echo "== SQ_SPAN_PROBE=$v (-1: everything; 2: DMA alone; 1: counting alone)"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
for r in sorted(rows, key=lambda r: r['Name']):
    if 'k_span<' in r['Name']: print('  ', r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')"
  rm -rf $OUT/st$v
done | tee $OUT/summary.txt
