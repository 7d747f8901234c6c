#!/bin/bash
# round 5, after the register drop (no sorted build spills any more): reads of many lengths, one wave for both streams against a
# wave per stream (SQ_SPAN_SORTED_SPLIT=1), the bench's ragged configuration; one line per run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_sorted_split
mkdir -p $OUT
: > $OUT/summary.txt
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" $OUT/$name.json >> $OUT/summary.txt <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().split("\n")[-1])
except Exception as e:
    print(f"{name:34s} FAILED ({e})"); raise SystemExit
for k, v in (d.get("other_configs") or {}).items():
    if not isinstance(v, dict) or "roofline" not in v: continue
    print(f"{name:34s} {k:26s} {v['value']:9.1f} Gbases/s  frac {v['roofline']['frac']:.4f}  {v.get('route', '')}")
PY
}
for i in 1 2; do
run ragged_default_$i                     -- --configs ragged_50_150
run ragged_sorted_split_$i SQ_SPAN_SORTED_SPLIT=1 -- --configs ragged_50_150
done
run ragged_waves12 SQ_SPAN_SORTED_SPLIT=1 SQ_SPAN_WAVES=12 -- --configs ragged_50_150
cat $OUT/summary.txt
