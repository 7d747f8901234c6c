#!/bin/bash
# A/B on ONE box: scripts/build/libsqgpu_before.so (SQ_LIB) against the tree's library, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ab
for i in 1 2 3; do
  for v in before new; do
    if [ $v = before ]; then export SQ_LIB=$PWD/scripts/build/libsqgpu_before.so; else unset SQ_LIB; fi
    timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --configs uniform_200bp,config3_paired_by_tile > gpurun_out/r5ab/$v$i.json 2>/dev/null
    python - $v gpurun_out/r5ab/$v$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], "headline", d["value"], d["roofline"]["avg_launch_ms"], " ".join(f"{k} {v['value']}" for k, v in d["other_configs"].items()))
PY
  done
done
