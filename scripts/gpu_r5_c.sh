#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r5c/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5c/tests.log; grep "^FAILED\|^ERROR" gpurun_out/r5c/tests.log | head -40
bash scripts/exp_w6.sh
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --configs config3_paired,config3_paired_by_tile,config3_paired_five_calls_unfused > gpurun_out/r5c/c3.json 2> gpurun_out/r5c/c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/c3.json").read().strip().split("\n")[-1])
for k, v in d["other_configs"].items():
    print(k, v["value"], v["roofline"]["frac"], v.get("route"), v.get("checks"))
PY
