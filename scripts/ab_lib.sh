#!/bin/bash
# same-box A/B of experiment builds (scripts/build_exp.sh): the headline's launch time, three rounds.  usage: scripts/ab_lib.sh NAME [NAME ...]
cd "${GRAFT_REPO_ROOT:?}" || exit 1
if [ -n "$TESTS" ]; then for n in "$@"; do SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 600 python -m pytest $TESTS -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | sed "s/^/$n: /"; done; fi
for i in 1 2 3; do
  for n in "$@"; do
    if [ -n "$LEN" ]; then SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 300 python scripts/bench_len.py $LEN ${READS:-25000000} 2>&1 | grep "AdapterCounter" | sed "s/^/$n /"; continue; fi
    SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-other-configs ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', d['value'], d['roofline']['avg_launch_ms'], d.get('route'))"
  done
done
