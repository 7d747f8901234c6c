#!/usr/bin/env python3
"""Experiment: does the fused pass speed up when every 32-byte chunk window of a read is
32-byte aligned in memory?  Records are laid out name|seq|qual without separators
(the FastqRecordArrayView([views]) layout) so that alignment is under control."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth
from sequali_amd._qc import META_DTYPE, FastqRecordArrayView

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(1)


def layout(nl, L, lead=0):
    rec = nl + 2 * L
    buf = np.empty(lead + n * rec + 64, dtype=np.uint8)
    body = buf[lead:lead + n * rec].reshape(n, rec)
    body[:, :nl] = ord("x")
    body[:, nl:nl + L] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L))
    body[:, nl + L:] = rng.integers(35, 74, size=(n, L), dtype=np.uint8)
    metas = np.zeros(n, dtype=META_DTYPE)
    metas["record_start"] = lead + np.arange(n, dtype=np.uint64) * rec
    metas["name_length"] = nl
    metas["sequence_offset"] = nl
    metas["sequence_length"] = L
    metas["qualities_offset"] = nl + L
    metas["tags_offset"] = nl + 2 * L
    return FastqRecordArrayView._from_buffer(buf.tobytes(), metas)


def timed(label, arr, ad):
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)) if ad else None)
    arr._device()
    f.add_record_array(arr); f.qc_metrics._pending.clear(); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f.add_record_array(arr); f.qc_metrics._pending.clear()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / 5
    bases = len(arr) * int(arr._metas["sequence_length"][0])
    print(f"{label:50s} {dt*1e3:8.2f} ms {bases/dt/1e9:8.1f} Gbases/s", flush=True)


for ad in (True, False):
    os.environ["SQ_NO_RING"] = "1"   # k_pass for both, to see the effect of the alignment alone
    print("QC+AD" if ad else "QC only (k_pass)")
    timed("L=160 nl=32 rec=352: all windows 32B aligned", layout(32, 160), ad)
    timed("L=160 nl=33 rec=353: rotating alignment", layout(33, 160), ad)
    timed("L=160 nl=32 lead=16: all windows straddle 32B", layout(32, 160, 16), ad)
    timed("L=192 nl=64 rec=448: all windows 64B-pair aligned", layout(64, 192), ad)
    timed("L=192 nl=65 rec=449: rotating alignment", layout(65, 192), ad)
