#!/bin/bash
# round 3: PerTileQuality's pass over the headers on a stream of its own (default) or on the work stream (SQ_PT_PREP_INLINE=1)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3r
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_golden.py tests/test_gpu_staging.py tests/test_gpu_reference_suite.py tests/test_gpu_shards.py -q -m gpu -x -k "tile or ptq or pertile or config3 or golden or staging or pair" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | tee $OUT/summary.txt; grep -B30 "^E " $OUT/tests.log | head -60 | tee -a $OUT/summary.txt
for v in "SQ_X=0" "SQ_PT_PREP_INLINE=1" "SQ_X=0" "SQ_PT_PREP_INLINE=1"; do
  env $v python scripts/bench_config3.py | tail -1 | sed "s/^/$v /"
done | tee -a $OUT/summary.txt
