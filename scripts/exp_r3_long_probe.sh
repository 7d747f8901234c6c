#!/bin/bash
# round 3: where the cycles of k_span<LONG> go (probe library: scripts/build/libsqgpu_probe.so), config 4
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3p
mkdir -p $OUT
cd $R
cp scripts/build/libsqgpu_probe.so sequali_amd/libsqgpu.so
for v in "SQ_LONG_NW=8" "SQ_LONG_NW=6" "SQ_LONG_NW=4" "SQ_LONG_NW=8 SQ_SPAN_SPILLS_OK=1"; do
echo "== $v"
env $v SQ_SPAN_STAMPS=1 python bench.py --kind nanopore --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -A1 "stamps per span" | tail -2
done | tee $OUT/summary.txt
