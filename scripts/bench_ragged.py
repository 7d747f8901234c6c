#!/usr/bin/env python3
"""Reads of many lengths (what adapter trimming leaves): the bench's 150 bp records with every
read cut to a random length in [lo, 150] by editing the metas; QCMetrics + AdapterCounter fused.
python scripts/bench_ragged.py [reads] [lo]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics, _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = synth.device_array(synth.ILLUMINA, 0, n)
buf, metas = dev._batch.download()
rng = np.random.default_rng(5)
metas = metas.copy()
metas["sequence_length"] = rng.integers(lo, 151, size=n).astype(metas["sequence_length"].dtype)
arr = FastqRecordArrayView._from_buffer(buf, metas)
bases = int(metas["sequence_length"].sum())
f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
for label, a, nb in (("uniform 150 bp", dev, 150 * n), (f"lengths {lo}..150", arr, bases)):
    f.add_record_array(a); f.qc_metrics._pending.clear(); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f.add_record_array(a); f.qc_metrics._pending.clear()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{label:22s} {dt * 1e3:8.2f} ms  {nb / dt / 1e9:8.1f} Gbases/s")
