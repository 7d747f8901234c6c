#!/bin/bash
# round 5: the two decisions about k_wide<AD>: (a) adapters of 14-25 characters: k_wide (default) against the six-dword
# builds of k_span (SQ_SPAN_W6=1) at 100, 150, 200, 224 bases; (b) 225-256 bases with 12-mers: k_wide against
# k_span<8,AD,uniform,split> (SQ_SPAN_NW8=1 was an experiment hook in sq_span_launch, removed after this measurement)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_w6
mkdir -p $OUT
: > $OUT/summary.txt
for L in 100 150 200 224; do
  echo "== L=$L 20-mers, default" >> $OUT/summary.txt
  SQ_BENCH_PROBES=long timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
  echo "== L=$L 20-mers, SQ_SPAN_W6=1" >> $OUT/summary.txt
  SQ_BENCH_PROBES=long SQ_SPAN_W6=1 timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
for L in 240 250 256; do
  echo "== L=$L 12-mers, default" >> $OUT/summary.txt
  SQ_BENCH_PROBES=short timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
  echo "== L=$L 12-mers, SQ_SPAN_NW8=1" >> $OUT/summary.txt
  SQ_BENCH_PROBES=short SQ_SPAN_NW8=1 timeout 300 python scripts/bench_len.py $L 4000000 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
cat $OUT/summary.txt
