#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
scripts/build/ubench_qsad > gpurun_out/r5e/ubench_qsad.txt 2>&1; cat gpurun_out/r5e/ubench_qsad.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r5e/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5e/tests.log; grep "^FAILED\|^ERROR" gpurun_out/r5e/tests.log | head -40
SQ_BENCH_PROBES=short timeout 300 python scripts/bench_len.py 250 4000000 2>&1 | grep -v amdgpu.ids
