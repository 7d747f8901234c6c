#!/bin/bash
# round 3: timing builds (-DSQ_SPAN_PROBE, results wrong by construction) and counters of k_span<5,true>,
# one wave for both streams (SQ_SPAN_SPLIT=0) and a wave per stream (1)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT
cd $R
# counters first, on the product library
for sp in 0 1; do
  KERNEL=k_span TAG=r3b_split$sp SETS="a b" ARGS="--reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 --no-other-configs" bash scripts/pmc.sh SQ_SPAN_SPLIT=$sp > /dev/null 2>&1
  cp $R/gpurun_out/pmc_r3b_split$sp/summary.txt $OUT/pmc_split$sp.txt
  cd /tmp; export TMPDIR=/tmp
  SQ_SPAN_SPLIT=$sp rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAVES --kernel-trace --output-format csv -d $OUT/fifo$sp -- python3 $R/bench.py --reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 --no-other-configs > $OUT/fifo$sp.log 2>&1
  python3 - $OUT/fifo$sp <<'PY' >> $OUT/pmc_split$sp.txt
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        if "k_span" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    for c, v in sorted(acc.items()): print(f"k_span {c:28s} {v / cnt[c]:18.0f} ({cnt[c]} launches)")
PY
  cd $R
done
cp sequali_amd/libsqgpu.so $OUT/libsqgpu_product.so
trap 'cp $OUT/libsqgpu_product.so $R/sequali_amd/libsqgpu.so; rm -f $OUT/*.so $OUT/*.o' EXIT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wno-unused-function -DSQ_SPAN_PROBE -DSQ_SPAN_ONLY_NW=5"
hipcc $F -c sequali_amd/csrc/sq_qc.hip -o $OUT/sq_qc.o &
hipcc $F -c sequali_amd/csrc/sq_span.hip -o $OUT/sq_span.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o sequali_amd/libsqgpu.so sequali_amd/build/sq_api.o $OUT/sq_qc.o $OUT/sq_span.o sequali_amd/build/sq_ends.o sequali_amd/build/sq_nano.o || exit 1
B="python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs"
for sp in 0 1; do
  for m in 0 1 2 3 4 8 16 24; do
    if [ $sp = 0 ] && [ $m -ge 8 ]; then continue; fi
    SQ_SPAN_SPLIT=$sp SQ_SPAN_PROBE=$m $B 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $sp probe $m (1: no DMA, 2: no counting, 3: neither, 4: DMA into a slot nobody reads, 8: bases not counted, 16: qualities not counted): %.3f ms per launch' % d['roofline']['avg_launch_ms'])"
  done
  SQ_SPAN_SPLIT=$sp SQ_SPAN_STAMPS=1 $B --steps 1 --warmup 0 2>&1 | grep "stamps per span" | tail -1
done | tee $OUT/summary.txt
