import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import OverrepresentedSequences, _lib, synth
n = 10_000_000
r1 = synth.device_array(synth.ILLUMINA, 0, n)
ov = OverrepresentedSequences()
for _ in range(4):
    t0 = time.perf_counter(); ov.add_record_array(r1); _lib.synchronize(); print(f"{(time.perf_counter()-t0)*1e3:.2f} ms", ov.collected_unique_fragments)
