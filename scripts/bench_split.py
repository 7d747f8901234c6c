"""Throughput of the GPU FASTQ record split (sq_batch_from_fastq_device) on text that
is already resident in HBM, next to the host memchr splitter (sq_fastq_split)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sequali_amd import synth  # noqa: E402
from sequali_amd._lib import context, lib  # noqa: E402
from sequali_amd._qc import META_DTYPE, _DeviceBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
arr = synth.device_array(0, 0, n)
hip = C.CDLL("libamdhip64.so")
nbytes = lib().sq_batch_bytes(arr._batch.handle)
d_text = lib().sq_batch_device_text(arr._batch.handle)
consumed = C.c_size_t(0)
for rep in range(4):
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    h = lib().sq_batch_from_fastq_device(context(), C.c_void_p(d_text), nbytes, C.byref(consumed))
    hip.hipDeviceSynchronize()
    dt = time.perf_counter() - t0
    assert h and consumed.value == nbytes and lib().sq_batch_size(h) == n
    lib().sq_batch_free(h)
    print(f"device split: {n} records, {nbytes/1e9:.2f} GB in {dt*1e3:.2f} ms = {nbytes/dt/1e9:.1f} GB/s, {n*150/dt/1e9:.1f} Gbases/s")
m = min(n, 2_000_000)
text, _ = synth.host_records(0, 0, m)
buf = np.frombuffer(text, np.uint8)
metas = np.zeros(m, META_DTYPE)
t0 = time.perf_counter()
got = lib().sq_fastq_split(buf.ctypes.data, len(buf), metas.ctypes.data, m, C.byref(consumed))
dt = time.perf_counter() - t0
print(f"host split: {got} records in {dt*1e3:.1f} ms = {len(buf)/dt/1e9:.2f} GB/s (1 core)")
