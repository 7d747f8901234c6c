#!/bin/bash
# full GPU test-suite, default bench line, rocprofv3 kernel stats of the same command -> gpurun_out/round/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-900
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --cpu-sample 0 > $OUT/bench_profiled.json 2> $OUT/prof.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -5 $OUT/kernel_stats.csv | cut -c1-200
