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


def _plumbing_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sequali_amd.dist import _all_reduce, _broadcast_bytes, all_gather_ragged
    mine = torch.arange(3 + 4 * rank, dtype=torch.int64) + 100 * rank          # ragged lengths
    keys = torch.full((rank + 1, 32), rank + 1, dtype=torch.uint8)              # [n, 32] rows
    got = all_gather_ragged(mine)
    got_keys = all_gather_ragged(keys)
    empty = all_gather_ragged(torch.zeros(0 if rank == 0 else 2, dtype=torch.int64))
    state = _broadcast_bytes(b"state of rank 1 \x00\xff" if rank == 1 else b"", 1, None)
    total = _all_reduce(torch.tensor([rank + 1, 10], dtype=torch.int64))
    low = _all_reduce(torch.tensor([5 - rank]), dist.ReduceOp.MIN)
    np.savez(os.path.join(out_dir, f"plumbing{rank}.npz"), got=got.numpy(), keys=got_keys.numpy(),
             empty=empty.numpy(), state=np.frombuffer(state, np.uint8), total=total.numpy(), low=low.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_merge_plumbing_two_ranks(tmp_path):
    """the collectives the order-dependent merges are built from (dist.merge_*): ragged
    all-gather in rank order, byte broadcast for the estimator relay, reductions"""
    world = 2
    mp.spawn(_plumbing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        g = np.load(tmp_path / f"plumbing{rank}.npz")
        assert g["got"].tolist() == [0, 1, 2, 100, 101, 102, 103, 104, 105, 106]
        assert g["keys"].shape == (3, 32) and g["keys"][:, 0].tolist() == [1, 2, 2]
        assert g["empty"].tolist() == [0, 0]
        assert g["state"].tobytes() == b"state of rank 1 \x00\xff"
        assert g["total"].tolist() == [3, 20] and g["low"].tolist() == [4]
