#!/bin/bash
# round 3: where a span's cycles go (probe build with s_memtime stamps between the phases of a span)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
cd $R
cp sequali_amd/libsqgpu.so $OUT/libsqgpu_product.so
trap 'cp $OUT/libsqgpu_product.so $R/sequali_amd/libsqgpu.so; rm -f $OUT/*.so $OUT/*.o' EXIT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-function -DSQ_SPAN_PROBE -DSQ_SPAN_ONLY_NW=5 $EXTRA_FLAGS"
hipcc $F -c sequali_amd/csrc/sq_qc.hip -o $OUT/sq_qc.o &
hipcc $F -c sequali_amd/csrc/sq_span.hip -o $OUT/sq_span.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o sequali_amd/libsqgpu.so sequali_amd/build/sq_api.o $OUT/sq_qc.o $OUT/sq_span.o sequali_amd/build/sq_ends.o sequali_amd/build/sq_nano.o sequali_amd/build/sq_feed.o || exit 1
B="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-other-configs"
for sp in 0 1; do
  echo "== split $sp"
  SQ_SPAN_SPLIT=$sp SQ_SPAN_STAMPS=1 $B 2>&1 | grep -A1 "stamps per span" | tail -2
  echo "== split $sp, QCMetrics alone"
  SQ_SPAN_SPLIT=$sp SQ_SPAN_STAMPS=1 $B --modules qc 2>&1 | grep -A1 "stamps per span" | tail -2
done | tee $OUT/summary.txt
