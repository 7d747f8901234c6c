#!/bin/bash
# round 5, after the register drop: adapters of 20 characters, k_wide (SQ_SPAN_W6=0) against the six-dword builds (SQ_SPAN_W6=1)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_w6b
mkdir -p $OUT
: > $OUT/summary.txt
for L in 80 100 128 250; do
  for w in 0 1; do
    echo "== L=$L 20-mers, SQ_SPAN_W6=$w" >> $OUT/summary.txt
    SQ_BENCH_PROBES=long SQ_SPAN_W6=$w timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
  done
done
cat $OUT/summary.txt
timeout 300 python -m pytest tests/test_gpu_span_edges.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
