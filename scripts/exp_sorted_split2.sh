#!/bin/bash
# round 5, after the class codes' fast path: the length-sorted route once more, one wave for both streams (SQ_SPAN_SORTED_SPLIT=0)
# against a wave per stream (1), with and without adapters, three alternations
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_sorted_split
mkdir -p $OUT
: > $OUT/summary2.txt
for i in 1 2 3; do for s in 0 1; do
  SQ_SPAN_SORTED_SPLIT=$s timeout 300 python scripts/bench_ragged_dev.py 12500000 50 2>&1 | grep "reads of" | sed "s/^/SORTED_SPLIT=$s  /" >> $OUT/summary2.txt
done; done
cut -c1-230 $OUT/summary2.txt
