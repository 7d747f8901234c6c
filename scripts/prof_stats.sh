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
if [ -n "$TRACE_KERNEL" ]; then   # every launch of the kernels whose name holds $TRACE_KERNEL, in order: ms
  find "$OUT/st_$tag" -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open('{}')) if '$TRACE_KERNEL' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print('$TRACE_KERNEL launches (ms):', ' '.join('%.2f' % ((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows))"
fi
rm -rf "$OUT/st_$tag"
cut -c1-200 "$OUT/kernel_stats_$tag.csv" | head -${LINES_SHOWN:-25}
