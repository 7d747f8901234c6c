#!/bin/bash
# k_uni vs k_pass (SQ_UNI=0), QCMetrics + AdapterCounter, 25 M reads per launch:
# FETCH_SIZE, where the waves wait, LDS activity
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/uni_pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="--reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 $EXTRA"
for mode in ${MODES:-uni pass}; do
  unset SQ_UNI SQ_WIDE; if [ $mode = pass ]; then export SQ_UNI=0; fi; if [ $mode = wide ]; then export SQ_WIDE=1; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${mode}_f -- python3 $R/bench.py $ARGS > $OUT/${mode}_f.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/${mode}_a -- python3 $R/bench.py $ARGS > $OUT/${mode}_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/${mode}_b -- python3 $R/bench.py $ARGS > $OUT/${mode}_b.log 2>&1
done
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for mode in ("uni", "pass", "wide"):
    for run in "fab":
        for f in glob.glob(f"{mode}_{run}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "k_uni" in k or "k_pass" in k or "k_ring" in k or "k_wide" in k:
                    k = k[28:60]
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
            for k, d in acc.items():
                for c, v in sorted(d.items()):
                    print(f"{mode:5s} {k:34s} {c:24s} {v / cnt[(k, c)]:18.0f}")
PY
