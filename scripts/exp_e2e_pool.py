#!/usr/bin/env python3
"""the default path with all six modules: seconds in sq_batch_free and the pool's hipMalloc / hipFree calls per pass
python scripts/exp_e2e_pool.py [reads]"""
import ctypes as C
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import (AdapterCounter, DedupEstimator, FastqParser, FusedPass, NanoStats, OverrepresentedSequences,  # noqa: E402
                         PerTileQuality, QCMetrics, _lib, _qc, synth)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
text = synth.illumina_fastq(0, n)
free_s = [0.0, 0]
real_free = _qc._DeviceBatch.__del__


def timed_free(self):
    t = time.perf_counter()
    real_free(self)
    free_s[0] += time.perf_counter() - t
    free_s[1] += 1


_qc._DeviceBatch.__del__ = timed_free


def counts():
    out = (C.c_uint64 * 4)()
    _lib.lib().sq_pool_counts(_lib.context(), out)
    return list(out)


def run(six):
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality() if six else None)
    more = (OverrepresentedSequences(), NanoStats(), DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)) if six else ()
    free_s[0], free_s[1] = 0.0, 0
    c0 = counts()
    t0 = time.perf_counter()
    for a in FastqParser(io.BytesIO(text)):
        f.add_record_array(a)
        for mod in more:
            mod.add_record_array(a)
    t1 = time.perf_counter()
    f.qc_metrics.flush()
    for mod in more:
        mod.flush()
    _lib.synchronize()
    t2 = time.perf_counter()
    c1 = counts()
    print(f"{'six' if six else 'two'}: loop {1e3 * (t1 - t0):.1f} ms, flush {1e3 * (t2 - t1):.1f} ms; {free_s[1]} frees {1e3 * free_s[0]:.2f} ms; "
          f"pool mallocs +{c1[0] - c0[0]} frees +{c1[1] - c0[1]} idle {c1[2]} waiting {c1[3]}")


for six in (True, True, True, True, False, False, False):
    run(six)
