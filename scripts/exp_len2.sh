#!/bin/bash
# round 5: with the builds' registers down by a third, more waves per CU for 161-256 bases: k_span<6..8,AD,uniform,split> again
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_len2
mkdir -p $OUT
: > $OUT/summary.txt
for L in 176 200 224; do
  echo "== L=$L 12-mers" >> $OUT/summary.txt
  SQ_BENCH_PROBES=short timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
for L in 240 250 256; do
  echo "== L=$L 12-mers, default (k_wide)" >> $OUT/summary.txt
  SQ_BENCH_PROBES=short timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
  echo "== L=$L 12-mers, SQ_SPAN_NW8=1 (k_span<8>, 12 waves)" >> $OUT/summary.txt
  SQ_BENCH_PROBES=short SQ_SPAN_NW8=1 timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
cat $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_routes.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
