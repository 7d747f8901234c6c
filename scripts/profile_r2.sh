#!/bin/bash
# Round-2 evidence (run on the GPU box): kernel stats of the default bench.py, FETCH_SIZE / WRITE_SIZE of
# its headline kernel, the same counters on a linear stream of known size (calibration) -> gpurun_out/r2b/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --cpu-sample 0 > $OUT/bench_profiled.json 2> $OUT/prof.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --reads 50000000 --steps 1 --warmup 1 --cpu-sample 0 --no-other-configs > $OUT/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/scripts/build/ubench_flat 8 > $OUT/cal_$c.log 2>&1
done
cd $OUT
python3 - <<'PY' | tee fetch_calibration.txt
import csv, glob, collections
def avg(pattern, kernel):
    out = {}
    for f in glob.glob(pattern, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            out[k] = (sum(v) / len(v), len(v))
    return out
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = avg(f"cal_{c}/**/*counter_collection.csv", "k_flat")
    hot = avg(f"pmc_{c}/**/*counter_collection.csv", "k_span")
    for name, d in (("k_flat (8 GiB linear stream, 8589934592 bytes read once)", cal), ("k_span<5,true> (25 M reads per launch, 8.7 GB algorithmic)", hot)):
        for k, (v, n) in d.items():
            print(f"{name}: {k} = {v:.0f} KB per launch ({n} launches) = {v * 1024 / 1e9:.3f} GB")
PY
# traffic.json of this build: FETCH_SIZE x 2 (the calibration above: a linear stream of N bytes shows N / 2) + WRITE_SIZE
python3 - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
vals = {}
for line in open("fetch_calibration.txt"):
    m = re.match(r"(k_flat|k_span).*: (\w+) = (\d+) KB per launch \((\d+) launches\)", line)
    if m: vals[(m.group(1), m.group(2))] = (int(m.group(3)), int(m.group(4)))
f, nf = vals[("k_span", "FETCH_SIZE")]; w, nw = vals[("k_span", "WRITE_SIZE")]
cal = vals[("k_flat", "FETCH_SIZE")][0] * 1024 / 8589934592.0
json.dump({"kind": "illumina", "modules": ["adapter", "qc"], "reads_per_launch": 25000000, "kernel": "k_span<5,true>",
           "csrc_sha": bench.csrc_sha(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": min(nf, nw),
           "fetch_size_of_a_linear_stream_per_byte_read": round(cal, 4),
           "hbm_bytes_per_launch": int(f * 1024 / cal + w * 1024), "algorithmic_bytes_per_launch": 8700000000,
           "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py --reads 50000000 --steps 1 --warmup 1 --no-other-configs; "
                   "FETCH_SIZE divided by what the same counter shows per byte of an 8 GiB linear stream (scripts/ubench_flat.hip, "
                   "0.5 on gfx950: the guide's x2 correction); WRITE_SIZE as reported; scripts/profile_r2.sh"},
          open("traffic.json", "w"), indent=1)
print(open("traffic.json").read())
PY
rm -rf pmc_FETCH_SIZE pmc_WRITE_SIZE cal_FETCH_SIZE cal_WRITE_SIZE
# the headline kernel alone (every k_span launch is one of the bench's 25 M-read launches)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_h -- python3 $R/bench.py --cpu-sample 0 --no-other-configs > $OUT/bench_profiled_headline.json 2> $OUT/prof_h.err
find $OUT/stats_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_headline.csv
rm -rf $OUT/stats_h
