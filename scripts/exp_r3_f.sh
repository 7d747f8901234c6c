#!/bin/bash
OUT=gpurun_out/r3f
mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs"
run() { echo "== $1" | tee -a $OUT/summary.txt; shift; env "$@" $B 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], r['avg_launch_ms'], r['frac'], d['checks'])" | tee -a $OUT/summary.txt; }
run "split 16 waves" SQ_SPAN_SPLIT=1
run "unsplit 12 waves" SQ_SPAN_SPLIT=0
run "split 12 waves" SQ_SPAN_SPLIT=1 SQ_SPAN_WAVES=12
echo "== qc only" | tee -a $OUT/summary.txt
for sp in 1 0; do SQ_SPAN_SPLIT=$sp $B --modules qc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('split=$sp', d['value'], r['avg_launch_ms'], r['frac'], d['checks'])" | tee -a $OUT/summary.txt; done
python scripts/dbg_overrep4.py 2>&1 | tail -20 | tee -a $OUT/summary.txt
