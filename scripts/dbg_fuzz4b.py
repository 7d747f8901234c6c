#!/usr/bin/env python3
"""round 5: the order in which the staged work of scripts/fuzz.py 5 55 4 runs (NanoStats saw error rates of 0.0)"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
import sequali_amd._qc as Q
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, InsertSizeMetrics, NanoStats, PerTileQuality, QCMetrics
rng = np.random.default_rng(3)
n, U = int(os.environ.get("DBG_N", "300")), int(os.environ.get("DBG_U", "200"))
def mk(tag):
    names = [f"read{i} ch={i % 512} start_time=2021-09-30T11:34:{i % 60:02d}Z" for i in range(n)]
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=U).tobytes().decode() for _ in range(n)]
    quals = [(rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode() for _ in range(n)]
    return oracle.make_batch(names, seqs, quals)
b1, m1 = mk(1); b2, m2 = mk(2)
for name in ("FusedPass", "NanoStats", "InsertSizeMetrics", "QCMetrics"):
    cls = getattr(Q, name)
    def wrap(orig, name):
        def _run(self, arr, *a):
            print(f"   RUN {name} on array of {len(arr)} records, device metas at {Q.lib().sq_batch_device_metas(arr._device().handle):#x}", flush=True)
            return orig(self, arr, *a)
        return _run
    if hasattr(cls, "_run"): cls._run = wrap(cls._run, name)
    if hasattr(cls, "_run_pair"):
        def wrap2(orig, name):
            def _run_pair(self, a1, a2):
                print(f"   RUN_PAIR {name} {len(a1)}", flush=True)
                return orig(self, a1, a2)
            return _run_pair
        cls._run_pair = wrap2(cls._run_pair, name)
cuts = [int(x) for x in os.environ.get("DBG_CUTS", "0,100,220,300").split(",")]
for with_pairs in (False, True):
    print("with_pairs", with_pairs, flush=True)
    rq, rn = oracle.QCMetrics(), oracle.NanoStats()
    q, a, p, z, ns = QCMetrics(), AdapterCounter(["AGATCGGAAGAG"]), PerTileQuality(), InsertSizeMetrics(), NanoStats()
    f = FusedPass(q, a, p)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            x1 = m1[lo:hi].copy()
            rq.add(b1, x1); rn.add(b1, x1)
            a1 = FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy())
            a2 = FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy())
            f.add_record_array(a1)
            if with_pairs: z.add_record_array_pair(a1, a2)
            ns.add_record_array(a1)
        print(" getters: q", flush=True); q.number_of_reads
        print(" getters: z", flush=True); z.insert_sizes()
        print(" getters: n", flush=True)
        gi, ri = ns.nano_infos(), rn.nano_infos()
    bad = np.nonzero(gi["cumulative_error_rate"].view(np.uint64) != ri["cumulative_error_rate"].view(np.uint64))[0]
    print(" nanostats differ:", len(bad), bad[:5], flush=True)
