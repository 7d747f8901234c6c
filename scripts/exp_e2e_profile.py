#!/usr/bin/env python3
"""cProfile of the default path with all six modules (FastqParser at 128 KiB over a BytesIO, one call per module and array):
where the host's time goes.  python scripts/exp_e2e_profile.py [reads]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import (AdapterCounter, DedupEstimator, FastqParser, FusedPass, NanoStats, OverrepresentedSequences,  # noqa: E402
                         PerTileQuality, QCMetrics, _lib, synth)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
text = synth.illumina_fastq(0, n)


def run():
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality())
    more = (OverrepresentedSequences(), NanoStats(), DedupEstimator(front_sequence_offset=64, back_sequence_offset=0))
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
    print(f"loop {1e3 * (t1 - t0):.1f} ms, flush {1e3 * (t2 - t1):.1f} ms")


run()
run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
