#!/bin/bash
# Round 5, after the parity steps (scripts/gpu_bisect_r5.sh all) are green: every opt-in of round 4 beside its default, one
# gpurun call, one line per run in gpurun_out/exp_r5/summary.txt:
#   gpurun --timeout 1500 -- 'bash scripts/exp_r5.sh'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_r5
mkdir -p $OUT
run() {   # name, env assignments ..., -- , bench arguments
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
    print(f"{name:44s} FAILED ({e})"); raise SystemExit
rows = [("headline", d)] if "--no-other" in name else []
rows += list((d.get("other_configs") or {}).items())
for k, v in rows:
    if not isinstance(v, dict) or "roofline" not in v: continue
    print(f"{name:34s} {k:36s} {v['value']:9.1f} Gbases/s  frac {v['roofline']['frac']:.4f}  {v.get('route', '')[:90]}  checks {all(v.get('checks', {}).values())}")
PY
}
: > $OUT/summary.txt
run headline--no-other                  -- --no-other-configs
run config4_default                     -- --configs config4_nanopore
# (round 5 also ran SQ_LONG_BLOCK=1024/4096/16384, SQ_LONG_OVERLAP=1/2/4 and SQ_SORTED_STREAMS=1 here: all lost and were deleted,
# profiles/r5/exp_opt_ins_summary.txt)
run ragged                              -- --configs ragged_50_150
run config3                             -- --configs config3_paired,config3_paired_by_tile,config3_paired_five_calls_unfused
run uniform_250_w6      SQ_SPAN_W6=1    -- --configs uniform_250bp,uniform_200bp
run uniform_250                         -- --configs uniform_250bp,uniform_200bp
cat $OUT/summary.txt
