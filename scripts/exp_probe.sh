#!/bin/bash
# How does the memory system take the fused pass's gather?  Builds libsqgpu with -DSQ_PROBE (load-only
# kernels in sq_qc.hip, selected by SQ_PROBE_MODE inside sq_fused_add_batch) into a scratch copy and
# times them on the bench's 25 M x 150 bp launches (8.7 GB of records per launch):
#   32 / 64 / 128: a wave visits its 64 rows that many bytes per row and stream at a time
#                  (2 / 4 / 8 lanes side by side on a row), the next visit in flight
#   33 / 65:       lane = row, 32 / 64 bytes per visit (every lane on a line of its own)
#   1:             the whole buffer as one linear stream;  2: wave-private spans of 64 records
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/probe
rm -rf $OUT; mkdir -p $OUT
cd $R
cp sequali_amd/libsqgpu.so $OUT/libsqgpu_product.so
trap 'cp $OUT/libsqgpu_product.so $R/sequali_amd/libsqgpu.so; rm -f $OUT/*.so $OUT/*.o' EXIT   # the product library comes back whatever happens
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-function"
hipcc $F -DSQ_PROBE -c sequali_amd/csrc/sq_qc.hip -o $OUT/sq_qc_probe.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o sequali_amd/libsqgpu.so sequali_amd/build/sq_api.o $OUT/sq_qc_probe.o sequali_amd/build/sq_span.o sequali_amd/build/sq_ends.o sequali_amd/build/sq_nano.o || exit 1
for m in 32 64 128 33 65 1 2; do
  SQ_PROBE_MODE=$m python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); ms=d['roofline']['avg_launch_ms']; print('probe $m: %.3f ms per launch = %.2f TB/s of records' % (ms, 8.7/ms))"
done | tee $OUT/summary.txt
