#!/bin/bash
# round 5, after the register drop: QCMetrics alone, one wave for both streams (SQ_SPAN_SPLIT_QC=0) against a wave per stream (1)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_split_qc
mkdir -p $OUT
: > $OUT/summary.txt
for L in 100 150 170 180 192 200 250; do
  for v in 0 1; do
    echo "== L=$L SQ_SPAN_SPLIT_QC=$v" >> $OUT/summary.txt
    SQ_SPAN_SPLIT_QC=$v timeout 300 python scripts/bench_len.py $L 8000000 2>&1 | grep "QCMetrics alone" >> $OUT/summary.txt
  done
done
for v in 0 1; do
  echo "== bench.py --modules qc (config 2's reads), SQ_SPAN_SPLIT_QC=$v" >> $OUT/summary.txt
  SQ_SPAN_SPLIT_QC=$v timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs --modules qc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['roofline']['frac'], d.get('route'))" >> $OUT/summary.txt
done
cat $OUT/summary.txt
