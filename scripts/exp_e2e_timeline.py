#!/usr/bin/env python3
"""Where the time of the pinned 64 MiB device-split path goes: host-side seconds per buffer inside the
parser (upload + split, two syncs) and inside add_record_array, with and without the upload ahead, plus
the plain H2D rate of the same pages.   python scripts/exp_e2e_timeline.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqParser, PinnedReader, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._qc import FusedPass  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402

n = 2_000_000
text = synth.illumina_fastq(0, n)
reader = PinnedReader(text)


def one(label):
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    reader._pos = 0
    _lib.synchronize()
    t0 = time.perf_counter()
    marks = []
    it = iter(FastqParser(reader, initial_buffersize=64 << 20, split_on_device=True))
    while True:
        a0 = time.perf_counter()
        try:
            arr = next(it)
        except StopIteration:
            break
        a1 = time.perf_counter()
        f.add_record_array(arr)
        a2 = time.perf_counter()
        del arr
        a3 = time.perf_counter()
        marks.append((a1 - a0, a2 - a1, a3 - a2))
    f.qc_metrics.flush()
    _lib.synchronize()
    dt = time.perf_counter() - t0
    m = np.array(marks) * 1e3
    print(f"{label}: {150 * n / dt / 1e9:.2f} Gbases/s, {dt * 1e3:.1f} ms, {len(text) / dt / 1e9:.1f} GB/s of text; per buffer (ms): "
          f"parser {np.round(m[:, 0], 2).tolist()} add {np.round(m[:, 1], 2).tolist()} free {np.round(m[:, 2], 2).tolist()}", flush=True)


for rep in range(3):
    one("ahead")
os.environ["SQ_AHEAD"] = "0"
for rep in range(3):
    one("no ahead")
# the bus alone: the same pages in 64 MiB pieces through hipMemcpyAsync on one stream
import torch  # noqa: E402
dev = torch.empty(len(text), dtype=torch.uint8, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for off in range(0, len(text), 64 << 20):
        m = min(64 << 20, len(text) - off)
        hip.hipMemcpy(dev.data_ptr() + off, reader._address + off, m, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"hipMemcpy H2D of the pages, 64 MiB pieces: {len(text) / dt / 1e9:.1f} GB/s", flush=True)
