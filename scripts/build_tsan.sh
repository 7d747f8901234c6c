#!/bin/bash
# The host side of libsqgpu.so under ThreadSanitizer: the feeder's workers, its walker and the caller's thread (csrc/sq_feed.hip).
# CPU only, as scripts/build_asan.sh.   scripts/build_tsan.sh && scripts/run_tsan_tests.sh
set -e
cd "$(dirname "$0")/../sequali_amd"
OUT=build/tsan
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -Wno-option-ignored -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fsanitize=thread -fno-omit-frame-pointer -Wno-unused-function"
pids=()
for f in sq_api sq_qc sq_span sq_span_w6 sq_pair sq_ends sq_nano sq_feed sq_dist; do
  /opt/rocm/bin/hipcc $FLAGS -c csrc/$f.hip -o $OUT/$f.o &
  pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c csrc/sq_hostsimd.cpp -o $OUT/sq_hostsimd.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -shared-libsan -o $OUT/libsqgpu_tsan.so $OUT/*.o
echo $PWD/$OUT/libsqgpu_tsan.so
