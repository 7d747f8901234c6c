import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import DedupEstimator, _lib, synth
n = 10_000_000
r1 = synth.device_array(synth.ILLUMINA, 0, n)
r2 = synth.device_array(synth.ILLUMINA_R2, 0, n)
d = DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)
dp = DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
for _ in range(4):
    t0 = time.perf_counter(); d.add_record_array(r1); _lib.synchronize(); t1 = time.perf_counter()
    dp.add_record_array_pair(r1, r2); _lib.synchronize(); t2 = time.perf_counter()
    print(f"single {(t1-t0)*1e3:.2f} ms (bits {d._modulo_bits})   paired {(t2-t1)*1e3:.2f} ms (bits {dp._modulo_bits})")
