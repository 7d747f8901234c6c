#!/usr/bin/env python3
"""Where the time of the default path (FastqParser at 128 KiB over host memory, one add_record_array per array) goes:
seconds inside readinto, inside sq_feeder_next (the record split), inside add_record_array, and the rest.
python scripts/exp_e2e_default_timeline.py [reads]"""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqParser, QCMetrics, _lib, synth  # noqa: E402
from sequali_amd._qc import FusedPass  # noqa: E402
from sequali_amd import _qc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
text = synth.illumina_fastq(0, n)


class Timed(io.BytesIO):
    spent = 0.0

    def readinto(self, b):
        t0 = time.perf_counter()
        r = io.BytesIO.readinto(self, b)
        Timed.spent += time.perf_counter() - t0
        return r


real_next = _lib.lib().sq_feeder_next
split = [0.0]


def timed_next(*a):
    t0 = time.perf_counter()
    r = real_next(*a)
    split[0] += time.perf_counter() - t0
    return r


for rep in range(4):
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    Timed.spent, split[0] = 0.0, 0.0
    fobj = Timed(text)
    _lib.lib().sq_feeder_next = timed_next
    _lib.synchronize()
    t0 = time.perf_counter()
    add = 0.0
    arrays = 0
    for a in FastqParser(fobj):
        t1 = time.perf_counter()
        f.add_record_array(a)
        add += time.perf_counter() - t1
        arrays += 1
    t2 = time.perf_counter()
    f.qc_metrics.flush()
    _lib.synchronize()
    t3 = time.perf_counter()
    _lib.lib().sq_feeder_next = real_next
    import ctypes as C
    dbg = (C.c_double * 4)()
    _lib.lib().sq_feeder_debug_times(dbg, 1)
    dt = t3 - t0
    print(f"   inside the feeder: new blocks {dbg[0] * 1e3:.1f} ms, record split {dbg[1] * 1e3:.1f} ms, fresh pinned allocations {dbg[2] * 1e3:.1f} ms ({int(dbg[3])} of them)")
    print(f"pass {rep}: {150 * n / dt / 1e9:.2f} Gbases/s, {dt * 1e3:.1f} ms for {len(text) / 1e6:.0f} MB in {arrays} arrays: readinto {Timed.spent * 1e3:.1f} ms, "
          f"sq_feeder_next {split[0] * 1e3:.1f} ms, add_record_array {add * 1e3:.1f} ms, final flush {1e3 * (t3 - t2):.1f} ms, "
          f"rest (iteration, objects) {1e3 * (dt - Timed.spent - split[0] - add - (t3 - t2)):.1f} ms", flush=True)
