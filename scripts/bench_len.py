#!/usr/bin/env python3
"""Uniform reads of a given length (random bases and qualities, the six Illumina probes planted in 5 % of the reads),
QCMetrics + AdapterCounter fused and QCMetrics alone: Gbases/s and the route taken.
python scripts/bench_len.py [length] [reads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._qc import META_DTYPE  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rng = np.random.default_rng(1)
name = 8
rec = 1 + name + 1 + L + 3 + L + 1
buf = np.empty((n, rec), dtype=np.uint8)
buf[:, 0] = ord("@"); buf[:, 1:1 + name] = ord("r"); buf[:, 1 + name] = 10
buf[:, 2 + name:2 + name + L] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L))
buf[:, 2 + name + L] = 10; buf[:, 3 + name + L] = ord("+"); buf[:, 4 + name + L] = 10
buf[:, 5 + name + L:5 + name + 2 * L] = rng.integers(35, 74, size=(n, L), dtype=np.uint8)
buf[:, -1] = 10
# SQ_BENCH_PROBES=long: the six probes with eight more letters each (20 characters: the adapter lengths k_span takes only
# in its six-dword builds, csrc/sq_span_w6.hip)
PROBES = list(synth.ILLUMINA_PROBES)
if os.environ.get("SQ_BENCH_PROBES") == "long":
    PROBES = [p + "ACGTTGCA"[i % 8:] + "ACGTTGCA"[:i % 8] for i, p in enumerate(PROBES)]
probe = np.frombuffer(PROBES[0].encode(), np.uint8)
hit = rng.random(n) < 0.05
at = rng.integers(0, L - len(probe), size=n)
for i in np.nonzero(hit)[0][:50000]:
    buf[i, 2 + name + at[i]:2 + name + at[i] + len(probe)] = probe
metas = np.zeros(n, dtype=META_DTYPE)
metas["record_start"] = np.arange(n, dtype=np.uint64) * rec + 1
metas["name_length"] = name
metas["sequence_offset"] = name + 1
metas["sequence_length"] = L
metas["qualities_offset"] = name + 1 + L + 3
metas["tags_offset"] = name + 1 + L + 3 + L
arr = FastqRecordArrayView._from_buffer(buf.tobytes(), metas)
for label, make in ((f"QCMetrics + AdapterCounter ({len(PROBES[0])}-mers)", lambda: FusedPass(QCMetrics(), AdapterCounter(PROBES))),
                    ("QCMetrics alone", lambda: FusedPass(QCMetrics(), None)))[:1 if os.environ.get("SQ_BENCH_PROBES") else 2]:
    f = make()
    f.add_record_array(arr); f.qc_metrics.flush(); _lib.synchronize()
    lib().sq_route_reset(context())
    t0 = time.perf_counter()
    for _ in range(3):
        f.add_record_array(arr); f.qc_metrics._pending.clear()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / 3
    route = (lib().sq_last_route(context()) or b"").decode().split("+")[0]
    print(f"{L} bases x {n} reads, {label}: {dt * 1e3:.3f} ms, {L * n / dt / 1e9:.1f} Gbases/s, {(2 * L + 48) * n / dt / 8e12:.3f} of 8 TB/s, route {route}", flush=True)
