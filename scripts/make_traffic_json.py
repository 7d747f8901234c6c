#!/usr/bin/env python3
"""profiles/traffic.json from what scripts/profile_configs.sh left in gpurun_out/r6/traffic_by_run.json (one bench.py run per
configuration; every run also holds ONE pass pair of the headline's kernel on a 25 M-read batch, whose bytes -- known from
the headline's own run -- are subtracted).  Several traffic_by_run.json files (a configuration profiled again) may be
given, later ones win:   python scripts/make_traffic_json.py [traffic_by_run.json ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

paths = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "r6", "traffic_by_run.json")]
runs, passes, sha, per_byte = {}, {}, None, None
for p in paths:
    d = json.load(open(p))
    if sha not in (None, d["csrc_sha"]):
        raise SystemExit(f"{p}: measured on other kernel sources ({d['csrc_sha']} != {sha})")
    sha, per_byte = d["csrc_sha"], d["fetch_size_of_a_linear_stream_per_byte_read"]
    runs.update(d["hbm_bytes_per_pass_by_run"])
    passes.update(d.get("passes_by_run", {}))
if sha != bench.csrc_sha():
    print(f"WARNING: measured on csrc {sha}, the tree holds {bench.csrc_sha()}: bench.py will not use these numbers", file=sys.stderr)
headline_per_launch = runs["headline"] // 2          # the headline's profile run: 50 M reads = two launches per pass
# (every run's headline batch was launched twice -- a warm-up and a step --, whatever the number of passes of the run's own entry)
others = {k: v - headline_per_launch * 2 // passes.get(k, 2) for k, v in runs.items() if k != "headline"}
out = {
    "kind": "illumina", "modules": ["adapter", "qc"], "reads_per_launch": 25000000,
    "kernel": "HEADLINE RUN: k_span<5, true, false, 3, true, false, false, 0>(PassParams, unsigned int)",
    "csrc_sha": sha, "fetch_size_of_a_linear_stream_per_byte_read": per_byte,
    "hbm_bytes_per_launch": headline_per_launch, "algorithmic_bytes_per_launch": 8700000000,
    "other_configs_hbm_bytes_per_step": others,
    "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), one bench.py run per configuration (--configs NAME --steps 1 --warmup 1, "
            "25 M-read batches): FETCH_SIZE divided by what the same counter shows per byte of an 8 GiB linear stream (scripts/ubench_flat.hip) "
            "+ WRITE_SIZE, summed over the run's kernels, / 2 passes, minus the headline kernel's one launch per pass that every run "
            "holds; scripts/profile_configs.sh + scripts/make_traffic_json.py; per-kernel values: profiles/r6/pmc_all.txt",
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
