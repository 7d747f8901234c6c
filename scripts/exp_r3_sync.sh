#!/bin/bash
# round 3: the two waves of a pair kept within a span of each other (SQ_SPAN_SYNC=1): speed and fetched bytes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3s
mkdir -p $OUT
cd $R
SQ_SPAN_SYNC=1 timeout 600 python -m pytest tests/test_gpu_vs_oracle.py tests/test_gpu_span_edges.py tests/test_gpu_golden.py -q -m gpu -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log | tee $OUT/summary.txt; grep -B30 "^E " $OUT/tests.log | head -60 | tee -a $OUT/summary.txt
for v in 0 1 0 1; do
SQ_SPAN_SYNC=$v timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('sync $v headline', d['value'], d['ms_per_step'], d['roofline']['frac'], all(d['checks'].values()))"
done | tee -a $OUT/summary.txt
for v in 0 1; do
SQ_SPAN_SYNC=$v timeout 300 python bench.py --kind nanopore --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sync $v nanopore', d['value'], d['ms_per_step'], all(d['checks'].values()))"
SQ_SPAN_SYNC=$v timeout 300 python scripts/bench_ragged.py 25000000 50 | tail -1 | sed "s/^/sync $v /"
done | tee -a $OUT/summary.txt
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  SQ_SPAN_SYNC=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f$v -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --reads 50000000 --no-other-configs > /dev/null 2>&1
  python3 -c "
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$OUT/f$v/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'k_span' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
print('sync $v FETCH_SIZE per launch (KB):', sum(v)/len(v), 'launches', len(v), ' x2 =', 2*sum(v)/len(v)*1024/1e9, 'GB')" | tee -a $OUT/summary.txt
  rm -rf $OUT/f$v
done
