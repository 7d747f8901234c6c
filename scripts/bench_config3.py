#!/usr/bin/env python3
"""Config 3 alone (what bench.py's other_configs.config3_paired runs): pairs of 150 bp records resident in HBM,
(QCMetrics + PerTileQuality) x 2 + InsertSizeMetrics.   python scripts/bench_config3.py [pairs] [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import FusedPass, InsertSizeMetrics, PerTileQuality, QCMetrics, _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
r1 = synth.device_array(synth.ILLUMINA, 0, n)
r2 = synth.device_array(synth.ILLUMINA_R2, 0, n)
fa, fb, isz = FusedPass(QCMetrics(), None, PerTileQuality()), FusedPass(QCMetrics(), None, PerTileQuality()), InsertSizeMetrics()


def step():
    fa.add_record_array(r1); fa.qc_metrics._pending.clear()
    fb.add_record_array(r2); fb.qc_metrics._pending.clear()
    isz.add_record_array_pair(r1, r2)


step()
_lib.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    step()
_lib.synchronize()
dt = (time.perf_counter() - t0) / passes
ok = fa.per_tile_quality.number_of_reads == n * (passes + 1) and isz.total_reads == n * (passes + 1)
print(f"config 3, {n} pairs: {dt * 1e3:.2f} ms per pass, {300 * n / dt / 1e9:.1f} Gbases/s, checks {ok}")
