#!/bin/bash
# PMC counters of one kernel of a bench.py run: every --pmc set in its own rocprofv3 run
# (--kernel-trace only).  Usage (on the GPU box):
#   KERNEL=k_span TAG=span [SETS="a b f w"] [ARGS="--reads 50000000 --steps 1 --warmup 1 --cpu-sample 0"] scripts/pmc.sh [ENV=VAL ...]
# Sets: a = where the waves wait, b = instruction counts and LDS, f = FETCH_SIZE, w = WRITE_SIZE.
# Writes gpurun_out/pmc_$TAG/summary.txt (averages per launch of the kernels whose name holds $KERNEL).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${TAG:-k}
rm -rf $OUT; mkdir -p $OUT
ARGS=${ARGS:-"--reads 50000000 --steps 1 --warmup 1 --cpu-sample 0"}
for kv in "$@"; do export "$kv"; done
for set in ${SETS:-a b}; do
  case $set in
    a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS";;
    b) C="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU";;
    f) C="FETCH_SIZE";;
    w) C="WRITE_SIZE";;
    c) C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY";;
    t) C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum";;
  esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$set -- python3 $R/bench.py $ARGS > $OUT/$set.log 2>&1
done
cd $OUT
KERNEL=${KERNEL:-k_} python3 - <<'PY' | tee summary.txt
import csv, glob, collections, os
want = os.environ["KERNEL"]
for f in sorted(glob.glob("*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if want in k:
            k = k.replace("(anonymous namespace)::", "")[:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            print(f"{k:42s} {c:24s} {v / cnt[(k, c)]:18.0f}   ({cnt[(k, c)]} launches)")
for f in sorted(glob.glob("*/**/*kernel_trace.csv", recursive=True))[:1]:
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            dur[row["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, v in dur.items():
        print(f"{k:42s} duration_ms (profiled)    {sum(v) / len(v):18.3f}   ({len(v)} launches)")
PY
