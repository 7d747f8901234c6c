#!/bin/bash
# Round-3 evidence (run on the GPU box): kernel stats of the default bench.py (all configs, and the
# headline alone), SQ counters and FETCH_SIZE / WRITE_SIZE of EVERY kernel of every config, the
# FETCH_SIZE calibration on a linear stream -> gpurun_out/r3/ (copy what is to be judged to profiles/r3/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --cpu-sample 0 > $OUT/bench_profiled.json 2> $OUT/prof.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_h -- python3 $R/bench.py --cpu-sample 0 --no-other-configs > $OUT/bench_profiled_headline.json 2> $OUT/prof_h.err
find $OUT/stats_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_headline.csv
rm -rf $OUT/stats_h
python3 $R/bench.py --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err
for set in a b f w; do
  case $set in
    a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS";;
    b) C="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES";;
    f) C="FETCH_SIZE";;
    w) C="WRITE_SIZE";;
  esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$set -- python3 $R/bench.py $ARGS > $OUT/pmc_$set.log 2>&1
  # the headline alone: every launch of its kernel is one of the bench's 25 M-read launches (the full run also sends the
  # end-to-end entries' small blocks through the same kernel)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/hpmc_$set -- python3 $R/bench.py $ARGS --reads 50000000 --no-other-configs > $OUT/hpmc_$set.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/scripts/build/ubench_flat 8 > $OUT/cal_$c.log 2>&1
done
cd $OUT
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench
def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
HEAD = "HEADLINE RUN: "
for f in glob.glob("pmc_*/**/*counter_collection.csv", recursive=True) + glob.glob("cal_*/**/*counter_collection.csv", recursive=True) + \
        glob.glob("hpmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = short(row["Kernel_Name"])
        if f.startswith("hpmc_"):
            if not name.startswith("k_span<"):
                continue
            name = HEAD + name
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("pmc_a/**/*kernel_trace.csv", recursive=True) + glob.glob("hpmc_a/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = short(row["Kernel_Name"])
        if f.startswith("hpmc_"):
            if not name.startswith("k_span<"):
                continue
            name = HEAD + name
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
calk = [k for k in acc if "k_flat" in k and acc[k].get("FETCH_SIZE")]
per_byte = (sum(acc[calk[0]]["FETCH_SIZE"]) / len(acc[calk[0]]["FETCH_SIZE"])) * 1024 / 8589934592.0 if calk else 0.5
with open("pmc_all.txt", "w") as out:
    out.write("averages per launch of every kernel of the default bench.py (--steps 1 --warmup 1), one rocprofv3 --pmc pass per counter set;\\n"
              "SQ_* cycle counters are quad-cycles summed over the waves; FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE of a linear stream = %.3f x its bytes)\\n\\n" % per_byte)
    for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
        if not any(x in k for x in ("k_", "DeviceRadixSort", "DeviceScan", "DeviceSelect")) or not any(acc[k].values()):
            continue
        d = acc[k]
        n = max(len(v) for v in d.values())
        ms = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        out.write(f"{k}   ({n} launches, {ms:.3f} ms avg under the profiler)\\n")
        for c in sorted(d):
            out.write(f"    {c:24s} {sum(d[c]) / len(d[c]):16.0f}\\n")
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            hbm = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 / per_byte + sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
            out.write(f"    hbm bytes per launch     {hbm:16.0f}   (FETCH_SIZE / {per_byte:.3f} + WRITE_SIZE)\\n")
        if "SQ_WAVE_CYCLES" in d and "SQ_WAVES" in d:
            pass
        out.write("\\n")
# HBM-side bytes per pass of every config: sum over the config's kernels of (bytes per launch x launches) / passes (2: one warm-up, one step)
def total(kernel_filter):
    t = 0.0
    for k, d in acc.items():
        if kernel_filter(k) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            t += sum(d["FETCH_SIZE"]) * 1024 / per_byte + sum(d["WRITE_SIZE"]) * 1024
    return t
uni = lambda k: k.startswith(HEAD + "k_span<5, true, false")
cfg = {
    "headline": (uni, 2 * 2),   # per LAUNCH: the headline run holds 50 M reads = 2 launches per pass
    "ragged_50_150": (lambda k: ("k_span<" in k and (", true, true, 3, true, false>" in k or ", true, true, 3, false, false>" in k)) or "k_span_scatter" in k or "k_span_keys" in k or "k_span_longer" in k or "DeviceRadixSort" in k, 2),
    "config3_paired": (lambda k: k.startswith("k_span<5, false, false") or any(x in k for x in ("k_ptspan", "k_tile_parse", "k_tile_assign", "k_isz_span", "k_isz_adapters", "k_tile_")), 2),
    "config4_nanopore": (lambda k: ", true, true>" in k or any(x in k for x in ("k_read_sums", "k_long_", "k_adapter_first", "k_stripe_counts")), 2),
}
res = {name: int(total(f) / div) for name, (f, div) in cfg.items()}
head = [k for k in acc if uni(k)]
tj = {"kind": "illumina", "modules": ["adapter", "qc"], "reads_per_launch": 25000000, "kernel": head[0] if head else "k_span<5,true,split>",
      "csrc_sha": bench.csrc_sha(), "fetch_size_of_a_linear_stream_per_byte_read": round(per_byte, 4),
      "hbm_bytes_per_launch": res["headline"], "algorithmic_bytes_per_launch": 8700000000,
      "other_configs_hbm_bytes_per_step": {k: v for k, v in res.items() if k != "headline"},
      "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the default bench.py --steps 1 --warmup 1 (all configs); FETCH_SIZE divided by what the "
              "same counter shows per byte of an 8 GiB linear stream (scripts/ubench_flat.hip); per config: sum over its kernels (by name) of bytes "
              "per launch x launches / passes; scripts/profile_r3.sh"}
json.dump(tj, open("traffic.json", "w"), indent=1)
print(open("traffic.json").read())
PY
rm -rf pmc_a pmc_b pmc_f pmc_w hpmc_a hpmc_b hpmc_f hpmc_w cal_FETCH_SIZE cal_WRITE_SIZE
