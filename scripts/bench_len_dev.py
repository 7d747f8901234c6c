#!/usr/bin/env python3
"""Batches of one read length generated in HBM (the benchmark generator cut / stretched to L: synth.with_length), QCMetrics +
AdapterCounter fused: Gbases/s and the route.  python scripts/bench_len_dev.py L [reads]   (round 5: k_wide against k_span below 129 bases)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25_000_000
arr = synth.device_array(synth.with_length(synth.ILLUMINA, L), 0, n)
f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
f.add_record_array(arr); f.qc_metrics.flush(); _lib.synchronize()
lib().sq_route_reset(context())
t0 = time.perf_counter()
for _ in range(4):
    f.add_record_array(arr); f.qc_metrics._pending.clear()
_lib.synchronize()
dt = (time.perf_counter() - t0) / 4
route = (lib().sq_last_route(context()) or b"").decode().split("+")[0]
print(f"{L} bases x {n} reads: {dt * 1e3:.3f} ms, {L * n / dt / 1e9:.1f} Gbases/s, {(2 * L + 48) * n / dt / 8e12:.3f} of 8 TB/s, route {route}", flush=True)
