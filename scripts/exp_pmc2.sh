#!/bin/bash
# where do the waves of the fused pass wait?  three --pmc passes over the default kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
ARGS="--reads 20000000 --batch-reads 10000000 --steps 1 --warmup 1 --cpu-sample 0 $EXTRA"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/c -- python3 $R/bench.py $ARGS > $OUT/c.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for run in "abc":
    for f in glob.glob(f"{run}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_ring" in k or "k_pass" in k:
                k = k[28:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
        for k, d in acc.items():
            for c, v in sorted(d.items()):
                print(f"{k:34s} {c:28s} {v / cnt[(k, c)]:18.0f}")
PY
