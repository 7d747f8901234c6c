"""N > 1 path on CPU: two gloo ranks shard a record array by contiguous ranges,
each accumulates its shard (with the CPU oracle standing in for the device
tables, which need a GPU), the tables are merged with sequali_amd.dist, and the
result equals the single-process tables."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pad(t: np.ndarray, rows: int, cols: int) -> np.ndarray:
    out = np.zeros(rows * cols, dtype=np.int64)
    out[:len(t)] = t.astype(np.int64)
    return out


def _worker(rank: int, world: int, port: int, n: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sequali_amd import synth
    from sequali_amd.dist import global_max, shard_range, sum_tables
    first, last = shard_range(n, rank, world)
    # ranks see reads of different maximum length: the merge has to pad
    kind = synth.NANOPORE
    buf, metas = synth.host_records(kind, first, last - first)
    m = oracle.QCMetrics()
    m.add(buf, metas)
    a = oracle.AdapterCounter(list(synth.NANOPORE_PROBES))
    a.add(buf, metas)
    ml = global_max(m.max_length)
    tables = [torch.from_numpy(_pad(m.base_count_table(), ml, 5)),
              torch.from_numpy(_pad(m.phred_count_table(), ml, 12)),
              torch.from_numpy(m.end_anchored_base_count_table().astype(np.int64)),
              torch.from_numpy(m.end_anchored_phred_count_table().astype(np.int64)),
              torch.from_numpy(m.gc_content().astype(np.int64)),
              torch.from_numpy(m.phred_scores().astype(np.int64))]
    for _, f, r in a.get_counts():
        tables.append(torch.from_numpy(_pad(f, ml, 1)))
        tables.append(torch.from_numpy(_pad(r, ml, 1)))
    reads = torch.tensor([m.number_of_reads])
    dist.all_reduce(reads)
    sum_tables(tables)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), ml=ml, reads=int(reads.item()),
                 **{f"t{i}": t.numpy() for i, t in enumerate(tables)})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from sequali_amd.dist import shard_range
    for total in (0, 1, 7, 100, 101):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_merge_equals_single_process(tmp_path):
    from sequali_amd import synth
    n, world = 60, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "merged.npz")
    buf, metas = synth.host_records(synth.NANOPORE, 0, n)
    m = oracle.QCMetrics()
    m.add(buf, metas)
    a = oracle.AdapterCounter(list(synth.NANOPORE_PROBES))
    a.add(buf, metas)
    assert int(got["ml"]) == m.max_length and int(got["reads"]) == n
    want = [m.base_count_table(), m.phred_count_table(), m.end_anchored_base_count_table(),
            m.end_anchored_phred_count_table(), m.gc_content(), m.phred_scores()]
    for _, f, r in a.get_counts():
        want += [f, r]
    for i, w in enumerate(want):
        np.testing.assert_array_equal(got[f"t{i}"].astype(np.uint64), w, err_msg=f"table {i}")
