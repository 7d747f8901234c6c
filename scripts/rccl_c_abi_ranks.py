#!/usr/bin/env python3
"""The job's exchange step through the C ABI alone (csrc/sq_dist.hip: RCCL, no torch), for the day a node with more than one
GPU is there: N processes, one per GPU; rank 0 hands the RCCL id to the others through a file; every rank counts its own
shard (records [rank * R, (rank + 1) * R) of the synthetic generator), then sq_qcmetrics_allreduce / sq_adaptercounter_
allreduce; every rank must hold the job's tables (checked against sums that are known without counting).
    python scripts/rccl_c_abi_ranks.py [ranks] [reads per rank]
NOT yet run: no such node has been available."""
import ctypes as C
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, reads, id_path, out):
    os.environ["SQ_DEVICE"] = str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import numpy as np
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth
    lib, ctx = _lib.lib(), _lib.context()
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        _lib.check(lib.sq_rccl_unique_id(buf))
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(id_path + ".tmp", id_path)
    while not os.path.exists(id_path):
        time.sleep(0.05)
    ident = (C.c_uint8 * 128).from_buffer_copy(open(id_path, "rb").read())
    comm = lib.sq_rccl_comm_init(ctx, world, ident, rank)
    assert comm, _lib.last_error()
    qc, ad = QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))
    dev = synth.device_array(synth.ILLUMINA, rank * reads, reads)
    FusedPass(qc, ad).add_record_array(dev)
    qc.flush()
    t0 = time.perf_counter()
    _lib.check(lib.sq_qcmetrics_allreduce(qc._h, comm))
    _lib.check(lib.sq_adaptercounter_allreduce(ad._h, comm))
    dt = time.perf_counter() - t0
    base = int(np.array(qc.base_count_table(), np.uint64).sum())
    ok = base == world * reads * 150 and qc.number_of_reads == world * reads and ad.number_of_sequences == world * reads
    lib.sq_rccl_comm_destroy(comm)
    out.put((rank, ok, base, dt))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    mp.set_start_method("spawn")
    id_path = os.path.join(tempfile.mkdtemp(), "rccl_id")
    out = mp.Queue()
    procs = [mp.Process(target=rank_main, args=(r, world, reads, id_path, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=600) for _ in procs)
    for p in procs:
        p.join()
    for rank, ok, base, dt in results:
        print(f"rank {rank}: job tables on this rank {'ok' if ok else 'WRONG'} (bases {base}), the two all-reduces {dt * 1e3:.2f} ms")
    sys.exit(0 if all(r[1] for r in results) else 1)
