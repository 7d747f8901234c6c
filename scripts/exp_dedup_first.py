import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import DedupEstimator, _lib, synth
n = 25_000_000
r1 = synth.device_array(synth.ILLUMINA, 0, n)
_lib.synchronize()
d = DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)
t0 = time.perf_counter(); d.add_record_array(r1); _lib.synchronize(); t1 = time.perf_counter()
d.add_record_array(r1); _lib.synchronize(); t2 = time.perf_counter()
print(f"fresh estimator, first 25M-read batch: {(t1-t0)*1e3:.1f} ms; second: {(t2-t1)*1e3:.1f} ms; bits {d._modulo_bits}")
