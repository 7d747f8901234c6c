#!/bin/bash
# round 3: the k_span tests and the headline number (bench.py without the other configurations), three runs
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3l
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_golden.py -q -m gpu -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | tee $OUT/summary.txt; grep -B30 "^E " $OUT/tests.log | head -80 >> $OUT/summary.txt
for i in 1 2 3; do
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('checks'))" | tee -a $OUT/summary.txt
done
