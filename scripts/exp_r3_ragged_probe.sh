#!/bin/bash
# round 3: where the cycles of the length-sorted k_span launches go (probe library built beforehand:
# scripts/build/libsqgpu_probe.so = the product sources with -DSQ_SPAN_PROBE), one wave for both streams
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3m
mkdir -p $OUT
cd $R
cp scripts/build/libsqgpu_probe.so sequali_amd/libsqgpu.so
for sp in 0 1; do
echo "== split $sp"
SQ_SPAN_SPLIT=$sp SQ_SPAN_STAMPS=1 python scripts/bench_ragged.py 25000000 50 2>&1 | grep -A1 "stamps per span\|Gbases" | grep -v "^--" | tail -40
done | tee $OUT/summary.txt
