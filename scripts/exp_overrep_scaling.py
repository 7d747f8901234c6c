import sys, time
sys.path.insert(0, '.')
import numpy as np
from sequali_amd import OverrepresentedSequences, synth, _lib
n, per = 100_000_000, 25_000_000
batches = [synth.device_array(synth.ILLUMINA, k * per, per) for k in range(n // per)]
def run(label, **kw):
    o = OverrepresentedSequences(**kw)
    for rep in range(3):
        _lib.synchronize(); t0 = time.perf_counter()
        for b in batches: o.add_record_array(b)
        o.flush(); _lib.synchronize()
        dt = time.perf_counter() - t0
        print(label, kw, f"pass {rep}: {dt*1e3:.2f} ms", "unique", o.collected_unique_fragments, flush=True)
run("default")
run("small table", max_unique_fragments=100_000)
run("tiny table", max_unique_fragments=1000)
run("every 16th", sample_every=16)
run("every 4th", sample_every=4)
run("ends 42+42", bases_from_start=42, bases_from_end=42)
