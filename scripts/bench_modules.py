#!/usr/bin/env python3
"""Per-module throughput on device-resident synthetic batches (informational;
bench.py is the contract benchmark).  python scripts/bench_modules.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import (AdapterCounter, DedupEstimator, FusedPass, InsertSizeMetrics,
                         OverrepresentedSequences, PerTileQuality, QCMetrics, _lib, synth)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
r1 = synth.device_array(synth.ILLUMINA, 0, n)
r2 = synth.device_array(synth.ILLUMINA_R2, 0, n)
bases = r1._batch.total_bases


def timed(label, fn, nbases, reps=3):
    fn(); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {dt * 1e3:9.2f} ms  {nbases / dt / 1e9:9.1f} Gbases/s", flush=True)


qc, ad, pt = QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality()
timed("QCMetrics", lambda: qc.add_record_array(r1) or qc._pending.clear(), bases)
timed("AdapterCounter (6 probes)", lambda: ad.add_record_array(r1), bases)
timed("PerTileQuality", lambda: pt.add_record_array(r1), bases)
f2 = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
timed("fused QCMetrics+AdapterCounter (config 2)", lambda: f2.add_record_array(r1) or f2.qc_metrics._pending.clear(), bases)
f3 = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality())
timed("fused QCMetrics+AdapterCounter+PerTileQuality", lambda: f3.add_record_array(r1) or f3.qc_metrics._pending.clear(), bases)
ov = OverrepresentedSequences()
timed("OverrepresentedSequences (1 in 8)", lambda: ov.add_record_array(r1), bases)
dd = DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)
timed("DedupEstimator single end", lambda: dd.add_record_array(r1), bases)
ddp = DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
timed("DedupEstimator paired", lambda: ddp.add_record_array_pair(r1, r2), 2 * bases)
isz = InsertSizeMetrics()
timed("InsertSizeMetrics", lambda: isz.add_record_array_pair(r1, r2), 2 * bases)
fa, fb = FusedPass(QCMetrics(), None, PerTileQuality()), FusedPass(QCMetrics(), None, PerTileQuality())
isz2 = InsertSizeMetrics()


def config3():
    fa.add_record_array(r1); fa.qc_metrics._pending.clear()
    fb.add_record_array(r2); fb.qc_metrics._pending.clear()
    isz2.add_record_array_pair(r1, r2)


timed("config 3: (QCMetrics+PerTileQuality) x2 + InsertSizeMetrics", config3, 2 * bases)
# the same reads in the order a sequencer writes them (tile by tile): no sort, no gather
rt = synth.device_array(synth.ILLUMINA_BY_TILE, 0, n)
pt2 = PerTileQuality()
timed("PerTileQuality, reads ordered by tile", lambda: pt2.add_record_array(rt), bases)
ft = FusedPass(QCMetrics(), None, PerTileQuality())
timed("QCMetrics+PerTileQuality, reads ordered by tile", lambda: ft.add_record_array(rt) or ft.qc_metrics._pending.clear(), bases)
print("dedup modulo bits", dd._modulo_bits, "tracked", dd.tracked_sequences,
      "| overrep unique", ov.collected_unique_fragments, "| insert sizes", sum(isz.insert_sizes()[1:]))
