#!/usr/bin/env python3
"""Reads of many lengths, made on the device (the bench's 150 bp records cut to lo..150 bases by sq_synth_trim, as bench.py's
ragged_50_150 does): QCMetrics + AdapterCounter fused and QCMetrics alone, time per pass and the route.
python scripts/bench_ragged_dev.py [reads] [lo] [read length: 150 or 250]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 50
kind = synth.ILLUMINA
dev = synth.device_array(kind, 0, n)
_lib.check(lib().sq_synth_trim(dev._batch.handle, 7, lo))
for label, make in (("QCMetrics + AdapterCounter", lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))),
                    ("QCMetrics alone", lambda: FusedPass(QCMetrics(), None))):
    f = make()
    f.add_record_array(dev); f.qc_metrics.flush(); _lib.synchronize()
    lib().sq_route_reset(context())
    t0 = time.perf_counter()
    for _ in range(4):
        f.add_record_array(dev); f.qc_metrics._pending.clear()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / 4
    route = (lib().sq_last_route(context()) or b"").decode()
    print(f"{n} reads of {lo}..150 bases, {label}: {dt * 1e3:.3f} ms per pass, route {route[:150]}", flush=True)
