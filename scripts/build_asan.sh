#!/bin/bash
# The host side of libsqgpu.so under AddressSanitizer + UBSan (the reference's own sanitizer run: tox.ini:68-76).
# CPU only: the sanitizers instrument the HOST half of every source (hipcc ignores -fsanitize for the gfx950 half, which
# is compiled as always and never launched by the tests that need no GPU), so the feeder's buffer arithmetic (csrc/sq_feed.hip), the vectorised newline scan
# (csrc/sq_hostsimd.cpp), the host record split and the automaton builders run instrumented.
#   scripts/build_asan.sh && scripts/run_asan_tests.sh
set -e
cd "$(dirname "$0")/../sequali_amd"
OUT=build/asan
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -Wno-option-ignored -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wno-unused-function"
pids=()
for f in sq_api sq_qc sq_span sq_span_w6 sq_pair sq_ends sq_nano sq_feed sq_dist; do
  /opt/rocm/bin/hipcc $FLAGS -c csrc/$f.hip -o $OUT/$f.o &
  pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c csrc/sq_hostsimd.cpp -o $OUT/sq_hostsimd.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libsqgpu_asan.so $OUT/*.o
echo $PWD/$OUT/libsqgpu_asan.so
