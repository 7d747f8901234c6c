#!/bin/bash
# round 5, second look: RCCL on one rank; config 3 on random tiles with the paired pass on; kernel stats of both config-3 entries
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5b
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_rccl.py -q -x -p no:cacheprovider > $OUT/rccl.log 2>&1; echo "rccl rc=$?"; tail -15 $OUT/rccl.log
SQ_PT_FUSED=1 timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --configs config3_paired,config3_paired_by_tile > $OUT/c3_fused.json 2> $OUT/c3_fused.err
python - $OUT/c3_fused.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
for k, v in d["other_configs"].items():
    print(k, v["value"], v["roofline"]["frac"], v.get("route"), v.get("checks"))
PY
cd /tmp
for cfg in config3_paired config3_paired_by_tile; do
  SQ_PT_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$cfg -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --configs $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  find $OUT/st_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_fused_$cfg.csv
  rm -rf $OUT/st_$cfg
  head -14 $OUT/kernel_stats_fused_$cfg.csv | cut -c1-200
done
