#!/bin/bash
# Timing builds of k_span (results wrong by construction): what the DMA alone and the counting alone cost.
# Builds libsqgpu with -DSQ_SPAN_PROBE into a scratch copy (the product library is restored by the trap).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/span_probe
rm -rf $OUT; mkdir -p $OUT
cd $R
cp sequali_amd/libsqgpu.so $OUT/libsqgpu_product.so
trap 'cp $OUT/libsqgpu_product.so $R/sequali_amd/libsqgpu.so; rm -f $OUT/*.so $OUT/*.o' EXIT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-function -DSQ_SPAN_PROBE"
hipcc $F -c sequali_amd/csrc/sq_qc.hip -o $OUT/sq_qc.o &
hipcc $F -c sequali_amd/csrc/sq_span.hip -o $OUT/sq_span.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o sequali_amd/libsqgpu.so sequali_amd/build/sq_api.o $OUT/sq_qc.o $OUT/sq_span.o sequali_amd/build/sq_ends.o sequali_amd/build/sq_nano.o || exit 1
for m in 0 1 2 3 4; do
  SQ_SPAN=1 SQ_SPAN_PROBE=$m python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('probe $m (1: no DMA, 2: no counting, 3: neither, 4: DMA into a slot nobody reads + counting on stale slots): %.3f ms per launch' % d['roofline']['avg_launch_ms'])"
done | tee $OUT/summary.txt
SQ_SPAN=1 SQ_SPAN_STAMPS=1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-other-configs 2>&1 | grep "stamps per span" | tail -2 | tee -a $OUT/summary.txt
