#!/bin/bash
# round 5, end: are the defaults of the older switches still the best on the final kernels?  one line per run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_r5_knobs
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
rows = [("headline", d)] if "headline" in name else []
rows += list((d.get("other_configs") or {}).items())
for k, v in rows:
    if not isinstance(v, dict) or "roofline" not in v: continue
    print(f"{name:34s} {k:26s} {v['value']:9.1f} Gbases/s  frac {v['roofline']['frac']:.4f}  {v.get('route', '')[:80]}")
PY
}
run headline_default                 -- --no-other-configs
run headline_waves12  SQ_SPAN_WAVES=12 -- --no-other-configs
run headline_nosync   SQ_SPAN_SYNC=0 -- --no-other-configs
run headline_unsplit  SQ_SPAN_SPLIT=0 -- --no-other-configs
run ragged_default                   -- --configs ragged_50_150
run ragged_sorted_split SQ_SPAN_SORTED_SPLIT=1 -- --configs ragged_50_150
run ragged_radix      SQ_SPAN_RADIX=1 -- --configs ragged_50_150
run config4_default                  -- --configs config4_nanopore
run config4_nw6       SQ_LONG_NW=6   -- --configs config4_nanopore
run config4_cost8     SQ_LONG_STRETCH_COST=8 -- --configs config4_nanopore
run config4_cost32    SQ_LONG_STRETCH_COST=32 -- --configs config4_nanopore
run config3_default                  -- --configs config3_paired,config3_paired_by_tile
run config3_tiles_only SQ_PT_FUSED=2 -- --configs config3_paired_by_tile
cat $OUT/summary.txt
