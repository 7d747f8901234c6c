#!/bin/bash
# round 3: k_span with the two-character automaton; split / unsplit; e2e host path
OUT=gpurun_out/r3d
mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs"
run() { echo "== $1" | tee -a $OUT/summary.txt; shift; env "$@" $B 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], r['avg_launch_ms'], r['frac'], d['checks'])" | tee -a $OUT/summary.txt; }
run "pair split 16 waves" SQ_SPAN_SPLIT=1
run "pair unsplit 12 waves" SQ_SPAN_SPLIT=0
run "pair split 12 waves" SQ_SPAN_SPLIT=1 SQ_SPAN_WAVES=12
