import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import InsertSizeMetrics, _lib, synth
n = 10_000_000
r1 = synth.device_array(synth.ILLUMINA, 0, n)
r2 = synth.device_array(synth.ILLUMINA_R2, 0, n)
z = InsertSizeMetrics()
for _ in range(3):
    t0 = time.perf_counter(); z.add_record_array_pair(r1, r2); _lib.synchronize(); print(f"{(time.perf_counter()-t0)*1e3:.2f} ms")
