#!/usr/bin/env python3
"""Config 3 alone (what bench.py's other_configs.config3_paired runs): pairs of 150 bp records resident in HBM,
(QCMetrics + PerTileQuality) x 2 + InsertSizeMetrics; reads in the order a sequencer writes (tile by tile) or with a
random tile each; per-module timings.   python scripts/bench_config3.py [pairs] [passes] [by_tile 0/1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import FusedPass, InsertSizeMetrics, PerTileQuality, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kinds = [int(sys.argv[3])] if len(sys.argv) > 3 else [1, 0]
for by_tile in kinds:
    r1 = synth.device_array(synth.ILLUMINA_BY_TILE if by_tile else synth.ILLUMINA, 0, n)
    r2 = synth.device_array(synth.ILLUMINA_R2_BY_TILE if by_tile else synth.ILLUMINA_R2, 0, n)
    fa, fb, isz = FusedPass(QCMetrics(), None, PerTileQuality()), FusedPass(QCMetrics(), None, PerTileQuality()), InsertSizeMetrics()

    def side_a():
        fa.add_record_array(r1); fa.qc_metrics._pending.clear()

    def side_b():
        fb.add_record_array(r2); fb.qc_metrics._pending.clear()

    def pairs():
        isz.add_record_array_pair(r1, r2)

    def step():
        side_a(); side_b(); pairs()

    lib().sq_route_reset(context())
    step()
    _lib.synchronize()
    route = (lib().sq_last_route(context()) or b"").decode()
    t0 = time.perf_counter()
    for _ in range(passes):
        step()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / passes
    parts = []
    for name, fn in (("read 1", side_a), ("read 2", side_b), ("insert sizes", pairs)):
        _lib.synchronize()
        t1 = time.perf_counter()
        for _ in range(passes):
            fn()
        _lib.synchronize()
        parts.append(f"{name} {(time.perf_counter() - t1) / passes * 1e3:.2f} ms")
    ok = fa.per_tile_quality.number_of_reads == n * (2 * passes + 1) and isz.total_reads == n * (2 * passes + 1)
    algo = (2 * 300 + 2 * (48 + 36)) * n
    print(f"config 3 ({'by tile' if by_tile else 'random tiles'}), {n} pairs: {dt * 1e3:.2f} ms per pass, {300 * n / dt / 1e9:.1f} Gbases/s, "
          f"{algo / dt / 8e12:.3f} of 8 TB/s; {', '.join(parts)}; checks {ok}; route {route}", flush=True)
    del r1, r2
