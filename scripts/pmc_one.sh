#!/bin/bash
# PMC counters of the kernels of one bench.py command whose name holds $KERNEL: every --pmc set in its own rocprofv3 run
# (--kernel-trace only) -> gpurun_out/r6/pmc_$TAG.txt (averages per launch).
# usage: KERNEL=k_overrep TAG=overrep scripts/pmc_one.sh [bench.py arguments ...]
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?}"
OUT=$R/gpurun_out/r6
mkdir -p "$OUT"
: > "$OUT/pmc_$TAG.txt"
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/pmc_${TAG}_run"
  rm -rf "$d"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$d" -- python3 "$R/bench.py" "$@" > /dev/null 2>&1
  python3 - "$d" "$KERNEL" >> "$OUT/pmc_$TAG.txt" <<'PY'
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k] / max(n[k], 1):18.1f}   per launch ({n[k]} launches)")
PY
  rm -rf "$d"
done
cat "$OUT/pmc_$TAG.txt"
