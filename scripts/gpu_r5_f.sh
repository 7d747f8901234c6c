#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5f
mkdir -p $OUT
cd $R
scripts/build/ubench_qsad 2>&1 | head -3
timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_routes.py "tests/test_gpu_vs_oracle.py::test_config3_one_million_pairs" tests/test_gpu_driver.py -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log; grep "^FAILED\|^ERROR" $OUT/tests.log | head
cd /tmp
for cfg in config3_paired config3_paired_by_tile; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$cfg -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --configs $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  find $OUT/st_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$cfg.csv
  rm -rf $OUT/st_$cfg
  head -9 $OUT/kernel_stats_$cfg.csv | cut -c1-200
  python3 - $OUT/bench_$cfg.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
for k, v in d["other_configs"].items():
    print(k, v["value"], v["roofline"]["frac"], v.get("route"), v.get("checks"))
PY
done
