#!/bin/bash
# same-box A/B of two libraries on config 4 (long reads): scripts/ab_config4.sh NAME NAME
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for n in "$@"; do
  SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --configs config4_nanopore 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['other_configs']['config4_nanopore']; print('$n', v['value'], v['roofline']['frac'], v.get('route'), v.get('checks'))"
done; done
