#!/usr/bin/env python3
"""the default path (FastqParser at 128 KiB over a BytesIO): the parse loop alone, with two and with six modules, for the
feeder's worker counts and with / without its walker.  python scripts/exp_feed_sweep.py [reads]"""
import io
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from sequali_amd import (AdapterCounter, DedupEstimator, FastqParser, FusedPass, NanoStats, OverrepresentedSequences,
                             PerTileQuality, QCMetrics, _lib, synth)
    text = synth.illumina_fastq(0, 2_000_000)

    def run(kind):
        f = more = None
        if kind != "parse":
            six = kind == "six"
            f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality() if six else None)
            more = (OverrepresentedSequences(), NanoStats(), DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)) if six else ()
        t0 = time.perf_counter()
        if f is None:
            n = 0
            for a in FastqParser(io.BytesIO(text)):
                n += 1
        else:
            for a in FastqParser(io.BytesIO(text)):
                f.add_record_array(a)
                for mod in more:
                    mod.add_record_array(a)
            f.qc_metrics.flush()
            for mod in more:
                mod.flush()
            _lib.synchronize()
        return 1e3 * (time.perf_counter() - t0)

    import ctypes as C
    out = []
    for kind in ("parse", "two", "six"):
        w = (C.c_double * 4)()
        run(kind)
        _lib.lib().sq_feeder_debug_waits(w, 1)
        times = sorted(run(kind) for _ in range(5))
        _lib.lib().sq_feeder_debug_waits(w, 1)
        out.append(f"{kind} {times[0]:.1f} / {times[2]:.1f} (per pass: waits for text {200 * w[0]:.1f} for the walker {200 * w[1]:.1f}, walker busy {200 * w[2]:.1f}, workers {200 * w[3]:.1f})")
    print(f"workers {os.environ.get('SQ_FEED_WORKERS', 'default')} walker {os.environ.get('SQ_FEED_WALKER', 'default')}: ms per 2 M reads (best / median of 5): " + ", ".join(out), flush=True)
else:
    for workers in (sys.argv[1:] or ["4"]):
        for walker in ("1", "0"):
            env = dict(os.environ, SQ_FEED_WORKERS=workers, SQ_FEED_WALKER=walker)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
