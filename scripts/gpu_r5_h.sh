#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5h
mkdir -p $OUT
for e in 0 1 2 3; do
  export SQ_EXP_ISZ=$e
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$e -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --configs config3_paired_by_tile > $OUT/bench_$e.json 2> $OUT/bench_$e.err
  python3 -c "
import csv,glob
for f in glob.glob('$OUT/st_$e/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_isz_adapters' in r['Name']: print('mode $e:', r['Calls'], float(r['AverageNs'])/1e6, 'ms')
"
  rm -rf $OUT/st_$e
done
