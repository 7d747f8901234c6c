#!/usr/bin/env python3
"""round 5: scripts/fuzz.py 200 55 failed in iteration 4 ('nanostats error rates'): which records, which call"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "0", "55"]
import importlib.util
spec = importlib.util.spec_from_file_location("fuzzmod", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz.py"))
src = open(spec.origin).read().split("ADS = [[")[0]      # the generator only
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz.py")}
exec(compile(src, "fuzz_head", "exec"), ns)
make, oracle = ns["make"], ns["oracle"]
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, NanoStats, PerTileQuality, QCMetrics, _lib
ADS = [["AGATCGGAAGAG", "CTGTCTCTTATA", "GGGGGGGGGGGG"], ["ACG", "NN", "GTAC", "TTTTTTTT"], ["ACGT" * 16, "A" * 40]]
it, seed0 = 4, 55
rng = np.random.default_rng(seed0 * 1000 + it)
n = int(rng.choice([1, 63, 64, 65, 500, 3000, 4500, 6000]))
max_len = int(rng.choice([5, 40, 151, 300, 700, 2500]))
if n * max_len > 16_000_000:
    n = 16_000_000 // max_len
uniform, illumina = bool(rng.random() < 0.4), bool(rng.random() < 0.7)
adapters = ADS[int(rng.integers(0, len(ADS)))]
cuts = sorted({0, n, *(int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3))))})
b1, m1 = make(rng, n, max_len, uniform, illumina, adapters)
print("n", n, "max_len", max_len, "uniform", uniform, "U", int(m1["sequence_length"][0]), "cuts", cuts, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
for variant in ("fused+nano", "fused, then flush, then nano", "qc alone + nano", "fused without pertile + nano"):
    rq, rn = oracle.QCMetrics(100), oracle.NanoStats()
    q, a, p, ns_ = QCMetrics(100), AdapterCounter(adapters), PerTileQuality(), NanoStats()
    f = FusedPass(q, a, p) if variant.startswith("fused") and "without" not in variant else (FusedPass(q, a) if "without" in variant else None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            x1 = m1[lo:hi].copy()
            rq.add(b1, x1); rn.add(b1, x1)
            a1 = FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy())
            if f is not None: f.add_record_array(a1)
            else: q.add_record_array(a1)
            if "then flush" in variant: q.flush()
            ns_.add_record_array(a1)
            got = a1.accumulated_error_rates()
            bad = np.nonzero(got.view(np.uint64) != x1["accumulated_error_rate"].view(np.uint64))[0]
            print(f"  {variant}: records [{lo}, {hi}): {len(bad)} error rates differ", bad[:5], got[bad[:3]], x1["accumulated_error_rate"][bad[:3]], "route", (_lib.lib().sq_last_route(_lib.context()) or b"").decode()[-120:], flush=True)
    gi, ri = ns_.nano_infos(), rn.nano_infos()
    badn = np.nonzero(gi["cumulative_error_rate"].view(np.uint64) != ri["cumulative_error_rate"].view(np.uint64))[0]
    print(f"{variant}: nanostats differ in {len(badn)} of {len(ri)} records, first {badn[:8]}", gi["cumulative_error_rate"][badn[:3]], ri["cumulative_error_rate"][badn[:3]], flush=True)
