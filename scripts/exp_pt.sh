#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pt
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $R/bench.py --modules ${MODS:-pertile} --reads 20000000 --batch-reads 10000000 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/log 2>&1
tail -1 $OUT/log | cut -c75-180
find $OUT/s -name "*kernel_stats.csv" | head -1 | xargs -I{} head -14 {} | cut -c1-150
