#!/bin/bash
# round 3: the device-side split without hipMalloc / hipFree per buffer and with the next buffer sent ahead
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd $R
python -m pytest tests/test_parser_golden.py tests/test_gpu_staging.py tests/test_gpu_driver.py -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | tee $OUT/summary.txt; grep -B30 "^E " $OUT/tests.log | head -80 >> $OUT/summary.txt
python scripts/exp_e2e_timeline.py 2>&1 | grep Gbases | cut -c1-400 | tee -a $OUT/summary.txt
