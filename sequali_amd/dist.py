"""Sharding and merging across ranks (one process per GPU).

The per-read accumulators are sums over independent records, so records are
sharded by contiguous ranges and the only exchange is one all-reduce of the
count tables: RCCL (backend "nccl") on device tensors that alias
the library's tables, gloo on CPU tensors in the tests.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) of the records rank `rank` takes; contiguous, sizes differ by <= 1"""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def global_max(value: int, device=None, group=None) -> int:
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def sum_tables(tables: Sequence[torch.Tensor], group=None) -> None:
    """In-place sum over ranks of equally shaped int64/float64 tensors, as ONE
    collective per dtype (the tables are KBs: latency bound, so fewer, larger)."""
    by_dtype = {}
    for t in tables:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape))
            off += n


class _Alias:
    """minimal __cuda_array_interface__ carrier for a raw device pointer"""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, count: int, device, typestr: str = "<i8") -> torch.Tensor:
    """torch view (no copy) of `count` 8-byte elements at device address `ptr`;
    u64 counters are viewed as int64 (sums below 2^63 have the same bits)"""
    return torch.as_tensor(_Alias(ptr, count, typestr), device=device)


def qcmetrics_tables(qc, device) -> List[torch.Tensor]:
    """device tensors over a QCMetrics object's six tables, trimmed to max_length"""
    from ._lib import lib
    ptrs = (ctypes.c_void_p * 8)()
    counts = (ctypes.c_uint64 * 8)()
    k = lib().sq_qcmetrics_device_tables(qc._h, ptrs, counts, 8)
    ml, ea = qc.max_length, qc.end_anchor_length
    want = [ml * 5, ml * 12, ea * 5, ea * 12, 101, 94]
    return [device_tensor(ptrs[i], want[i], device) for i in range(k) if ptrs[i] and want[i]]


def adaptercounter_tables(ad, device) -> List[torch.Tensor]:
    from ._lib import lib
    ptrs = (ctypes.c_void_p * 4)()
    counts = (ctypes.c_uint64 * 4)()
    k = lib().sq_adaptercounter_device_tables(ad._h, ptrs, counts, 4)
    return [device_tensor(ptrs[i], int(counts[i]), device) for i in range(k) if ptrs[i] and counts[i]]


def merge_qcmetrics(qc, device, group=None) -> None:
    """all ranks end up with the tables of the whole job"""
    from ._lib import check, lib, synchronize
    ml = global_max(qc.max_length, device, group)
    reads = torch.tensor([qc.number_of_reads], dtype=torch.int64, device=device)
    dist.all_reduce(reads, group=group)
    check(lib().sq_qcmetrics_set_totals(qc._h, qc.number_of_reads, ml))   # pads to ml rows
    synchronize()
    sum_tables(qcmetrics_tables(qc, device), group)
    torch.cuda.synchronize()
    check(lib().sq_qcmetrics_set_totals(qc._h, int(reads.item()), ml))


def merge_adaptercounter(ad, device, group=None) -> None:
    from ._lib import check, lib, synchronize
    ml = global_max(ad.max_length, device, group)
    seqs = torch.tensor([ad.number_of_sequences], dtype=torch.int64, device=device)
    dist.all_reduce(seqs, group=group)
    check(lib().sq_adaptercounter_set_totals(ad._h, ad.number_of_sequences, ml))
    synchronize()
    # every rank must present the same row length: reserve() may have over-allocated
    tables = adaptercounter_tables(ad, device)
    caps = torch.tensor([t.numel() for t in tables], dtype=torch.int64, device=device)
    lo = caps.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    if not torch.equal(lo, caps) or global_max(int(caps[0].item()), device, group) != int(caps[0].item()):
        raise RuntimeError("adapter tables differ in capacity across ranks; reserve() the same length first")
    sum_tables(tables, group)
    torch.cuda.synchronize()
    check(lib().sq_adaptercounter_set_totals(ad._h, int(seqs.item()), ml))
