#!/usr/bin/env python3
"""The route (sq_last_route) of QCMetrics + AdapterCounter, QCMetrics alone and QCMetrics + PerTileQuality for uniform
reads of many lengths and for a few ragged batches.  python scripts/routes_matrix.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics, synth  # noqa: E402
from sequali_amd._qc import META_DTYPE  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402


def batch(lengths):
    n = len(lengths)
    rng = np.random.default_rng(1)
    name = b"SIM:1:FCX:1:1101:5:7"
    parts, metas, at = [], np.zeros(n, dtype=META_DTYPE), 0
    for i, L in enumerate(lengths):
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes()
        rec = b"@" + name + b"\n" + seq + b"\n+\n" + b"I" * L + b"\n"
        metas[i] = (at + 1, len(name), len(name) + 1, L, len(name) + 1 + L + 3, len(name) + 1 + L + 3 + L, 0, 0.0)
        parts.append(rec)
        at += len(rec)
    return FastqRecordArrayView._from_buffer(b"".join(parts), metas)


def route(make, arr):
    f = make()
    lib().sq_route_reset(context())
    f.add_record_array(arr)
    f.qc_metrics.flush()
    return (lib().sq_last_route(context()) or b"").decode()


n = 8192
for L in (30, 64, 100, 150, 151, 160, 161, 192, 193, 224, 225, 256, 257, 300, 600):
    arr = batch([L] * n)
    print(f"uniform {L:4d}: AD {route(lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))), arr):40s} "
          f"QC {route(lambda: FusedPass(QCMetrics(), None), arr):30s} QC+PT {route(lambda: FusedPass(QCMetrics(), None, PerTileQuality()), arr)}")
rng = np.random.default_rng(2)
for lo, hi in ((50, 150), (20, 100), (100, 224), (100, 256), (150, 300)):
    arr = batch(rng.integers(lo, hi + 1, size=200_000).tolist())
    print(f"ragged {lo}..{hi}: AD {route(lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))), arr)}")
    print(f"ragged {lo}..{hi}: QC {route(lambda: FusedPass(QCMetrics(), None), arr)}")
