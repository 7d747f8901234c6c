#!/bin/bash
# the round-end checks on one box: pytest -m gpu, smoke(), the default bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/full
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log; grep -B30 "^E " $OUT/tests.log | head -60
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])
for k,v in d['other_configs'].items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('route'), v.get('checks'))"
