#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py command -> gpurun_out/r6/kernel_stats_TAG.csv (+ the bench line).
# usage: scripts/prof_stats.sh TAG [bench.py arguments ...]
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
OUT=$R/gpurun_out/r6
mkdir -p "$OUT"
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_$tag" -- python3 "$R/bench.py" "$@" > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
find "$OUT/st_$tag" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$tag.csv"
rm -rf "$OUT/st_$tag"
cut -c1-200 "$OUT/kernel_stats_$tag.csv" | head -${LINES_SHOWN:-25}
