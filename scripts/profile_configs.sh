#!/bin/bash
# The round's evidence (run on the GPU box; ROUND=r6 by default): kernel stats of the default bench.py and of the headline alone, then ONE
# configuration at a time (bench.py --configs NAME: several entries share kernels by name, so the counters of a
# configuration are those of a run that holds nothing else): kernel stats, FETCH_SIZE and WRITE_SIZE (each counter in its
# own --pmc pass), SQ counters for the headline and for config 3's passes -> gpurun_out/$ROUND/ (copy what is to be
# judged to profiles/$ROUND/).  Usage: [ROUND=r6] bash scripts/profile_configs.sh [stats|headline|CONFIG ...]   (default: everything; the
# whole script is ~ 25 GPU-minutes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${ROUND:-r6}
mkdir -p $OUT
WHAT="${@:-stats headline uniform_200bp uniform_250bp ragged_50_150 config3_paired config3_paired_by_tile config4_nanopore single_end_six_modules overrep_alone dedup_single_end dedup_paired insert_size_alone}"
SMALL="--steps 1 --warmup 1 --cpu-sample 0 --reads 25000000"
stats() {   # $1: tag, rest: bench.py arguments
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$tag -- python3 $R/bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  find $OUT/st_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/st_$tag
}
pmc() {     # $1: tag, $2: counters, rest: bench.py arguments
  tag=$1; C=$2; shift; shift
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py "$@" > $OUT/pmc_$tag.log 2>&1
}
for w in $WHAT; do
  case $w in
    stats)
      python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
      stats all --cpu-sample 0 ;;
    headline)
      stats headline --cpu-sample 0 --no-other-configs
      pmc headline_f FETCH_SIZE $SMALL --reads 50000000 --no-other-configs
      pmc headline_w WRITE_SIZE $SMALL --reads 50000000 --no-other-configs
      pmc headline_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $SMALL --reads 50000000 --no-other-configs
      pmc headline_b "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" $SMALL --reads 50000000 --no-other-configs ;;
    *)
      stats $w $SMALL --configs $w
      pmc ${w}_f FETCH_SIZE $SMALL --configs $w
      pmc ${w}_w WRITE_SIZE $SMALL --configs $w
      case $w in config3_paired)
        pmc ${w}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $SMALL --configs $w
        pmc ${w}_b "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" $SMALL --configs $w ;;
      esac ;;
  esac
done
for c in FETCH_SIZE WRITE_SIZE; do
  [ -x $R/scripts/build/ubench_flat ] && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/scripts/build/ubench_flat 8 > $OUT/cal_$c.log 2>&1
done
cd $OUT
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$R")
import bench
def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:72]
per_byte = 0.5
cal = collections.defaultdict(list)
for f in glob.glob("cal_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_flat" in row["Kernel_Name"]:
            cal[row["Counter_Name"]].append(float(row["Counter_Value"]))
if cal.get("FETCH_SIZE"):
    per_byte = sum(cal["FETCH_SIZE"]) / len(cal["FETCH_SIZE"]) * 1024 / 8589934592.0
tags = sorted({os.path.basename(d)[4:].rsplit("_", 1)[0] for d in glob.glob("pmc_*") if os.path.isdir(d)})
traffic, passes_of = {}, {}
with open("pmc_all.txt", "w") as out:
    out.write("per configuration (bench.py --configs NAME --steps 1 --warmup 1, 25 M-read batches): averages per launch of every kernel, one rocprofv3 --pmc pass per\\n"
              "counter set; FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE of a linear stream = %.3f x its bytes); hbm bytes = FETCH_SIZE / that + WRITE_SIZE\\n\\n" % per_byte)
    for tag in tags:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for f in [g for sfx in "fwab" for g in glob.glob(f"pmc_{tag}_{sfx}/**/*counter_collection.csv", recursive=True)]:   # (exactly this tag: config3_paired is a prefix of two others)
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for f in glob.glob(f"pmc_{tag}_f/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        out.write(f"==== {tag}\\n")
        total = 0.0
        for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
            d = acc[k]
            if not any(x in k for x in ("k_", "DeviceRadixSort", "DeviceScan", "DeviceSelect")) or "k_synth" in k or "k_batch_stats" in k:
                continue
            n = max(len(v) for v in d.values())
            ms = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
            out.write(f"{k}   ({n} launches, {ms:.3f} ms avg under the profiler)\\n")
            for c in sorted(d):
                out.write(f"    {c:24s} {sum(d[c]) / len(d[c]):16.0f}\\n")
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                hbm = sum(d["FETCH_SIZE"]) * 1024 / per_byte + sum(d["WRITE_SIZE"]) * 1024
                total += hbm
                out.write(f"    hbm bytes, all launches  {hbm:16.0f}\\n")
            out.write("\\n")
        passes = 3 if tag in ('single_end_six_modules', 'overrep_alone', 'dedup_single_end', 'dedup_paired') else 2   # one warm-up, one step; the settled modules: one more in front
        traffic[tag] = int(total / passes)
        passes_of[tag] = passes
        out.write(f"     {tag}: hbm bytes per pass over the run's records (everything above, / the run's passes): {traffic[tag]}\\n\\n")
json.dump({"csrc_sha": bench.csrc_sha(), "fetch_size_of_a_linear_stream_per_byte_read": round(per_byte, 4), "hbm_bytes_per_pass_by_run": traffic, "passes_by_run": passes_of,
           "note": "scripts/profile_configs.sh: one bench.py run per configuration (25 M-read batches; the headline's own batch is part of every run: subtract its "
                   "kernel's bytes, listed under each run in pmc_all.txt); to be turned into profiles/traffic.json by hand once the numbers have been looked at"},
          open("traffic_by_run.json", "w"), indent=1)
print(open("traffic_by_run.json").read())
PY
rm -rf pmc_*_f pmc_*_w pmc_*_a pmc_*_b cal_FETCH_SIZE cal_WRITE_SIZE
