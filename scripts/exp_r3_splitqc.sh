#!/bin/bash
# round 3: QCMetrics alone: one wave for both streams / a wave per stream (tied pairs); config 3 with both
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1 0 1; do
SQ_SPAN_SPLIT_QC=$v timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs --modules qc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('split_qc $v QCMetrics alone', d['value'], d['ms_per_step'], d['roofline']['frac'], all(d['checks'].values()))"
done
for v in 0 1 0 1; do
SQ_SPAN_SPLIT_QC=$v timeout 300 python scripts/bench_config3.py | tail -1 | sed "s/^/split_qc $v /"
done
