#!/usr/bin/env python3
"""the default path over a FILE on disk (the feeder's workers pread it) beside the same text in a BytesIO: ms per 2 M reads with
QCMetrics + AdapterCounter.  python scripts/exp_feed_file.py"""
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqParser, FusedPass, QCMetrics, _lib, synth  # noqa: E402

text = synth.illumina_fastq(0, 2_000_000)
path = os.path.join(tempfile.gettempdir(), "sq_exp_feed_file.fastq")
with open(path, "wb") as f:
    f.write(text)


def run(make):
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), None)
    t0 = time.perf_counter()
    src = make()
    for a in FastqParser(src):
        f.add_record_array(a)
    f.qc_metrics.flush()
    _lib.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    if hasattr(src, "close"):
        src.close()
    return dt


for name, make in (("BytesIO", lambda: io.BytesIO(text)), ("file (page cache)", lambda: open(path, "rb")),
                   ("file, SQ_FEEDER_SOURCE off (readinto)", None)):
    if make is None:
        from sequali_amd import _qc
        _qc._USE_SOURCE = False
        make = lambda: open(path, "rb")   # noqa: E731
    times = sorted(run(make) for _ in range(6))
    print(f"{name}: {times[0]:.1f} / {times[2]:.1f} ms per 2 M reads (best / median of 6) = {0.3 / times[2] * 1e3:.1f} Gbases/s")
os.unlink(path)
