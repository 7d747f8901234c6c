#!/bin/bash
# the paired-pass tests and config 3 timings
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pair
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_pair.py tests/test_gpu_routes.py tests/test_gpu_span_edges.py::test_few_very_long_reads_with_more_than_64_adapters "tests/test_gpu_vs_oracle.py::test_config3_one_million_pairs" -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log; grep -B5 -A25 "^E " $OUT/tests.log | head -120
timeout 600 python scripts/bench_config3.py > $OUT/c3.log 2>&1; tail -20 $OUT/c3.log
