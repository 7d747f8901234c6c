"""debug helper: one uniform-length batch through k_span, QC alone or with the automaton"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import os
U = int(sys.argv[1]); ad = sys.argv[2] == "ad"
os.environ["SQ_SPAN"] = "1"
from helpers import oracle
from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
rng = np.random.default_rng(1000 + U)
n = 64 * 5 + 37
probes = ["ACGTACGTACGT"[:min(U, 12)], "GGGGG"[:min(U, 5)], "TTNAC"[:min(U, 5)]]
names, seqs, quals = [], [], []
for i in range(n):
    s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=U, p=[.24, .24, .24, .24, .04]).tobytes().decode()
    names.append("r" * (1 + i % 67)); seqs.append(s)
    quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
buf, metas = oracle.make_batch(names, seqs, quals)
arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
gq, ga = QCMetrics(), AdapterCounter(probes)
if ad: FusedPass(gq, ga).add_record_array(arr)
else: gq.add_record_array(arr)
gq.flush()
print("ok", U, ad, int(np.asarray(gq.base_count_table()).sum()))
