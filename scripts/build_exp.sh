#!/bin/bash
# experiment builds: sq_span.hip again with extra flags (one window count: -DSQ_SPAN_ONLY_NW=5 unless NW=all), linked with the product's
# other objects into scripts/build/libsqgpu_$1.so.  usage: scripts/build_exp.sh NAME [flags ...]
name=$1; shift
only="-DSQ_SPAN_ONLY_NW=${NW:-5}"; [ "$NW" = all ] && only=""
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-function -Xclang -no-enable-noundef-analysis $only $*"
mkdir -p scripts/build
hipcc $F -c sequali_amd/csrc/sq_span.hip -o scripts/build/sq_span_$name.o || exit 1
B=sequali_amd/build
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/build/libsqgpu_$name.so $B/sq_api.o $B/sq_qc.o scripts/build/sq_span_$name.o $B/sq_ends.o $B/sq_nano.o $B/sq_feed.o $B/sq_pair.o $B/sq_span_w6.o $B/sq_dist.o $B/sq_hostsimd.o
