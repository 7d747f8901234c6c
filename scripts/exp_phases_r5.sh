#!/bin/bash
# round 5: where a span's cycles go on the final kernels (probe build of sq_span.hip with s_memtime stamps between the phases
# of a span: scripts/build/libsqgpu_probe.so, built with -DSQ_SPAN_PROBE from the same sources)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp_phases
mkdir -p $OUT
export SQ_LIB=$PWD/scripts/build/libsqgpu_probe.so SQ_SPAN_STAMPS=1
B="python bench.py --steps 1 --warmup 0 --cpu-sample 0"
{
echo "== headline"; $B --no-other-configs 2>&1 | grep -A1 "stamps per span" | tail -2
echo "== headline, QCMetrics alone"; $B --no-other-configs --modules qc 2>&1 | grep -A1 "stamps per span" | tail -2
echo "== headline, one wave for both streams"; SQ_SPAN_SPLIT=0 $B --no-other-configs 2>&1 | grep -A1 "stamps per span" | tail -2
echo "== ragged 50..150"; $B --configs ragged_50_150 2>&1 | grep -A1 "stamps per span" | tail -8
for L in 50 64 100 128 200 250; do
  echo "== one length: $L bases"; SQ_SPAN_SHORT=1 python scripts/bench_len.py $L 8000000 2>&1 | grep -A1 "stamps per span\|Gbases" | grep -v "^--" | tail -12
done
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
