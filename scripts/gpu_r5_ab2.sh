#!/bin/bash
# A/B on one box: copies of the automaton's table (SQ_SPAN_DFA_COPIES=1: one, as before; default: as many of 4, 2 as fit)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ab2
timeout 900 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_routes.py tests/test_gpu_golden.py tests/test_gpu_reference_suite.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r5ab2/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r5ab2/tests.log
for i in 1 2 3; do
  for v in 1 0; do
    SQ_SPAN_DFA_COPIES=$v timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --configs uniform_200bp,ragged_50_150,config4_nanopore > gpurun_out/r5ab2/c$v$i.json 2>/dev/null
    python - $v gpurun_out/r5ab2/c$v$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print("copies<=" + (sys.argv[1] if sys.argv[1] != "0" else "4"), "headline", d["value"], d["roofline"]["avg_launch_ms"], " ".join(f"{k} {v['value']}" for k, v in d["other_configs"].items()))
PY
  done
done
