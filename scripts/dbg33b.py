import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from sequali_amd import FastqRecordArrayView, QCMetrics
U = int(sys.argv[1]) if len(sys.argv) > 1 else 33
n = 64
names = ["r%d" % i for i in range(n)]
seqs = ["ACGT" * 20][0][:U]
seqs = [seqs] * n
quals = ["".join(chr(40 + (i + j) % 30) for j in range(U)) for i in range(n)]
buf, metas = oracle.make_batch(names, seqs, quals)
arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
q = QCMetrics()
q.add_record_array(arr)
try:
    q.flush()
    print("no error")
except Exception as e:
    print("error:", e)
rates = np.array(arr._metas["accumulated_error_rate"])
ref = oracle.QCMetrics(); ref.add(buf, metas)
exp = np.array(metas["accumulated_error_rate"])
print("nan at", np.nonzero(np.isnan(rates))[0][:20], "of", n)
if exp is not None:
    bad = np.nonzero(rates != exp)[0]
    print("differ at", bad[:20], rates[bad[:4]], exp[bad[:4]])
