#!/bin/bash
# FETCH_SIZE + kernel time of k_ring vs k_pass on 10 M reads/launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ring_pmc
mkdir -p $OUT
for mode in ring noring; do
  if [ $mode = noring ]; then export SQ_NO_RING=1; else unset SQ_NO_RING; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${mode}_fetch -- python3 $R/bench.py --reads 20000000 --batch-reads 10000000 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/${mode}_fetch.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/${mode}_sq -- python3 $R/bench.py --reads 20000000 --batch-reads 10000000 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/${mode}_sq.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_stats -- python3 $R/bench.py --reads 20000000 --batch-reads 10000000 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/${mode}_stats.log 2>&1
done
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for mode in ("ring", "noring"):
    for kind in ("fetch", "sq"):
        for f in glob.glob(f"{mode}_{kind}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"][:60]
                if "k_ring" in k or "k_pass" in k:
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[(k, row["Counter_Name"])] += 1
            for k, d in acc.items():
                print(mode, kind, k, {c: v / cnt[(k, c)] for c, v in d.items()})
    for f in glob.glob(f"{mode}_stats/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_ring" in row["Name"] or "k_pass" in row["Name"]:
                print(mode, "stats", row["Name"][:50], row["Calls"], row["AverageNs"])
PY
