#!/usr/bin/env python3
"""End to end from HOST memory (informational; bench.py is the contract benchmark, with the
records resident in HBM): FASTQ text in pinned host memory -> H2D copies on a copy stream,
double buffered -> record split on the GPU (sq_batch_from_fastq_device) -> fused QCMetrics +
AdapterCounter pass on the library's stream.  The copy of chunk i+1 overlaps the split and the
pass of chunk i.  Prints the PCIe-inclusive rate.   python scripts/bench_e2e.py [reads] [chunk_MiB]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._lib import check, context, lib  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
PREFIX = 1 << 20   # room in front of a chunk for the unfinished record of the chunk before

L = lib()
ctx = context()
dev = torch.device("cuda:0")
# ---- FASTQ text in pinned host memory ----
piece = 1_000_000
size = L.sq_synth_bytes(synth.ILLUMINA, synth.DEFAULT_SEED, 0, n_reads)
host = torch.empty(size, dtype=torch.uint8).pin_memory()
at = 0
for first in range(0, n_reads, piece):
    text = synth.illumina_fastq(first, min(piece, n_reads - first))
    host[at:at + len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8)
    at += len(text)
assert at == size
print(f"{n_reads} reads, {size / 1e9:.2f} GB of FASTQ text in pinned host memory", flush=True)

bufs = [torch.empty(PREFIX + chunk, dtype=torch.uint8, device=dev) for _ in range(2)]
copy_stream = torch.cuda.Stream()
work_stream = torch.cuda.ExternalStream(L.sq_stream_handle(ctx))
qc, ad = QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))


def run():
    offs = list(range(0, size, chunk))
    copied = [torch.cuda.Event() for _ in offs]
    with torch.cuda.stream(copy_stream):
        bufs[0][PREFIX:PREFIX + min(chunk, size)].copy_(host[0:min(chunk, size)], non_blocking=True)
        copied[0].record(copy_stream)
    rem = 0          # bytes of an unfinished record in front of the current chunk
    total = 0
    batches = []
    for i, off in enumerate(offs):
        cur = bufs[i % 2]
        n = min(chunk, size - off)
        if i + 1 < len(offs):   # next chunk's copy overlaps this chunk's work
            m = min(chunk, size - offs[i + 1])
            copy_stream.wait_stream(work_stream)   # the other buffer's last pass has finished
            with torch.cuda.stream(copy_stream):
                bufs[(i + 1) % 2][PREFIX:PREFIX + m].copy_(host[offs[i + 1]:offs[i + 1] + m], non_blocking=True)
                copied[i + 1].record(copy_stream)
        work_stream.wait_event(copied[i])
        start = PREFIX - rem
        consumed = C.c_size_t(0)
        h = L.sq_batch_from_fastq_device(ctx, C.c_void_p(cur.data_ptr() + start), rem + n, C.byref(consumed))
        if not h:
            raise RuntimeError(_lib.last_error())
        check(L.sq_fused_add_batch(h, qc._h, ad._h, None))
        total += L.sq_batch_size(h)
        batches.append(h)
        new_rem = rem + n - consumed.value
        if new_rem:
            with torch.cuda.stream(work_stream):
                nxt = bufs[(i + 1) % 2]
                nxt[PREFIX - new_rem:PREFIX].copy_(cur[start + consumed.value:start + rem + n], non_blocking=True)
        rem = new_rem
        if len(batches) > 1:   # the batch of the chunk before is done by now (its buffer is being refilled)
            L.sq_batch_free(batches.pop(0))
    _lib.synchronize()
    for h in batches:
        L.sq_batch_free(h)
    assert rem == 0 and total == n_reads, (rem, total)


run()  # warm up (allocations)
t0 = time.perf_counter()
run()
dt = time.perf_counter() - t0
print(f"end to end: {dt * 1e3:.1f} ms, {size / dt / 1e9:.1f} GB/s of FASTQ text over PCIe, "
      f"{n_reads * 150 / dt / 1e9:.1f} Gbases/s (split + QCMetrics + AdapterCounter included)")
assert qc.number_of_reads == 2 * n_reads
