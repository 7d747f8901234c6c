#!/bin/bash
# Runs the default bench step (no CPU baseline, no other configs) with every library variant
# scripts/build/libsqgpu_*.so in place of the product library (on the GPU box's scratch copy).
# Usage: [ENVS="SQ_SPAN=1"] [ARGS="--modules qc"] scripts/exp_lib_variants.sh
R=$GRAFT_REPO_ROOT
cd $R
cp sequali_amd/libsqgpu.so /tmp/libsqgpu_product.so
trap 'cp /tmp/libsqgpu_product.so $R/sequali_amd/libsqgpu.so' EXIT
for v in /tmp/libsqgpu_product.so scripts/build/libsqgpu_*.so; do
  cp $v sequali_amd/libsqgpu.so; touch sequali_amd/libsqgpu.so
  env $ENVS python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-other-configs $ARGS 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v: %.1f Gbases/s, %.3f ms per launch (%s)' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
done
