#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_routes.py tests/test_gpu_golden.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r5k/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r5k/tests.log
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --configs uniform_200bp,ragged_50_150,config4_nanopore,config3_paired_by_tile > gpurun_out/r5k/b$i.json 2>/dev/null; python - gpurun_out/r5k/b$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k, v in d["other_configs"].items(): print(" ", k, v["value"], v["roofline"]["frac"])
PY
done
