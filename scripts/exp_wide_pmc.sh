#!/bin/bash
# k_wide (default) vs k_pass (SQ_NO_WIDE=1), QCMetrics + AdapterCounter, 25 M reads per launch:
# FETCH_SIZE / WRITE_SIZE, where the waves wait, LDS activity.  Every --pmc set in its own run,
# --kernel-trace only.  Writes gpurun_out/wide_pmc/summary.txt and traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/wide_pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="--reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 $EXTRA"
for mode in ${MODES:-wide pass}; do
  unset SQ_NO_WIDE; if [ $mode = pass ]; then export SQ_NO_WIDE=1; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${mode}_f -- python3 $R/bench.py $ARGS > $OUT/${mode}_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${mode}_w -- python3 $R/bench.py $ARGS > $OUT/${mode}_w.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/${mode}_a -- python3 $R/bench.py $ARGS > $OUT/${mode}_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/${mode}_b -- python3 $R/bench.py $ARGS > $OUT/${mode}_b.log 2>&1
done
cd $OUT
python3 - <<'PY' | tee summary.txt
import csv, glob, collections, json
vals = {}
for mode in ("wide", "pass"):
    for run in "fwab":
        for f in glob.glob(f"{mode}_{run}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "k_wide" in k or "k_pass" in k:
                    k = k[28:60]
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
            for k, d in acc.items():
                for c, v in sorted(d.items()):
                    vals[(mode, c)] = v / cnt[(k, c)]
                    print(f"{mode:5s} {k:34s} {c:24s} {v / cnt[(k, c)]:18.0f}")
if ("wide", "FETCH_SIZE") in vals and ("wide", "WRITE_SIZE") in vals:
    f, w = vals[("wide", "FETCH_SIZE")], vals[("wide", "WRITE_SIZE")]
    json.dump({"kind": "illumina", "modules": ["adapter", "qc"], "reads_per_launch": 25000000,
               "kernel": "k_wide<AD>", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": 2,
               "hbm_bytes_per_launch": int((f + w) * 1024),
               "hbm_bytes_per_launch_upper_bound_x2_fetch": int((2 * f + w) * 1024),
               "algorithmic_bytes_per_launch": 8700000000,
               "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py --reads 50000000 --steps 1 --warmup 1 (scripts/exp_wide_pmc.sh); see profiles/README.md"},
              open("traffic.json", "w"), indent=1)
PY
