"""RCCL through the C ABI (csrc/sq_dist.hip) on ONE GPU: a communicator of one rank is a real
ncclCommInitRank, and the all-reduces of a one-rank job must leave every table as it is.  The many-rank run waits for
a node with more than one GPU (scripts/rccl_c_abi_ranks.py); what this pins is that librccl.so loads into the process,
that the entry points' argument order and enum values are RCCL's, and that the grouped in-place all-reduces of
QCMetrics' and AdapterCounter's tables run on the library's stream behind the pass that filled them.

In a process of its own, under a timeout: a communicator that does not come up must not hang the suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth
from oracle import oracle
lib, ctx = _lib.lib(), _lib.context()
ident = (C.c_uint8 * 128)()
_lib.check(lib.sq_rccl_unique_id(ident))
comm = lib.sq_rccl_comm_init(ctx, 1, ident, 0)
assert comm, _lib.last_error()
n = 20000
probes = list(synth.ILLUMINA_PROBES)
host = synth.host_array(synth.ILLUMINA, 0, n)
dev = synth.device_array(synth.ILLUMINA, 0, n)
qc, ad = QCMetrics(), AdapterCounter(probes)
FusedPass(qc, ad).add_record_array(dev)
qc.flush()
_lib.check(lib.sq_qcmetrics_allreduce(qc._h, comm))
_lib.check(lib.sq_adaptercounter_allreduce(ad._h, comm))
metas = host._metas.copy()
rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
rq.add(host.obj, metas); ra.add(host.obj, metas)
u64 = lambda x: np.array(x, dtype=np.uint64)
assert qc.number_of_reads == n and qc.max_length == rq.max_length and ad.number_of_sequences == n
np.testing.assert_array_equal(u64(qc.base_count_table()), rq.base_count_table())
np.testing.assert_array_equal(u64(qc.phred_count_table()), rq.phred_count_table())
np.testing.assert_array_equal(u64(qc.end_anchored_base_count_table()), rq.end_anchored_base_count_table())
np.testing.assert_array_equal(u64(qc.end_anchored_phred_count_table()), rq.end_anchored_phred_count_table())
np.testing.assert_array_equal(u64(qc.gc_content()), rq.gc_content())
np.testing.assert_array_equal(u64(qc.phred_scores()), rq.phred_scores())
hits = 0
for (_, f, r), (_, fr, rr) in zip(ad.get_counts(), ra.get_counts()):
    np.testing.assert_array_equal(u64(f), fr); np.testing.assert_array_equal(u64(r), rr)
    hits += int(u64(f).sum())
assert hits > 0
# the plain collectives: max of u64, sum of f64, all-gather of bytes
import torch
a = torch.arange(1000, dtype=torch.int64, device="cuda"); b = torch.linspace(0, 1, 777, dtype=torch.float64, device="cuda")
a0, b0 = a.clone(), b.clone()
torch.cuda.synchronize()
for op, t in ((2, a), (0, a), (1, b)):
    ptrs = (C.c_void_p * 1)(t.data_ptr()); counts = (C.c_uint64 * 1)(t.numel())
    _lib.check(lib.sq_rccl_allreduce_tables(ctx, comm, ptrs, counts, 1, op))
g = torch.zeros(1000 * 8, dtype=torch.uint8, device="cuda")
_lib.check(lib.sq_rccl_allgather_bytes(ctx, comm, C.c_void_p(a.data_ptr()), C.c_void_p(g.data_ptr()), 8000))
_lib.check(lib.sq_synchronize(ctx))
torch.cuda.synchronize()
assert torch.equal(a, a0) and torch.equal(b, b0) and torch.equal(g.view(torch.int64), a0)
lib.sq_rccl_comm_destroy(comm)
maps = open("/proc/self/maps").read()
assert "librccl" in maps
print("RCCL_ONE_RANK_OK")
''' % ROOT


def test_one_rank_communicator_and_both_allreduces():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired as e:
        raise AssertionError("a one-rank RCCL communicator did not come up in 240 s:\n" + str(e.stdout)[-2000:] + str(e.stderr)[-2000:])
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
