#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in 0 1; do
    SQ_SPAN_SORTED_SPLIT=$v timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --configs ragged_50_150 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); v=d['other_configs']['ragged_50_150']
print('sorted_split=$v', v['value'], v['roofline']['frac'], v['route'][:120])"
  done
done
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py -q -x -m gpu -p no:cacheprovider -k "sorted or ragged or trimmed" 2>&1 | tail -2
