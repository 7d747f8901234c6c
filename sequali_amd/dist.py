"""Sharding and merging across ranks (one process per GPU).

The per-read accumulators are sums over independent records, so records are
sharded by contiguous ranges and the only exchange is one all-reduce of the
count tables: RCCL (backend "nccl") on device tensors that alias
the library's tables, gloo on CPU tensors in the tests.

The modules with order-dependent state (first-come caps, the estimator's modulo
bits) are merged so that the job's result is the one a single sequential run over
all shards gives: see merge_overrepresented, merge_dedup, merge_insertsize,
merge_pertile.  Each takes the shard objects this process holds, in shard order
(normally one); with torch.distributed initialised the other ranks' shards join
through the collectives, without it the call merges the local shards only.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) of the records rank `rank` takes; contiguous, sizes differ by <= 1"""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def global_max(value: int, device=None, group=None) -> int:
    t = torch.tensor([value], dtype=torch.int64, device=device)
    return int(_all_reduce(t, dist.ReduceOp.MAX, group).item())


def sum_tables(tables: Sequence[torch.Tensor], group=None) -> None:
    """In-place sum over ranks of equally shaped int64/float64 tensors, as ONE
    collective per dtype (the tables are KBs: latency bound, so fewer, larger).
    RCCL reduces device tensors in place; gloo (tests, two ranks on one GPU) gets a host copy."""
    by_dtype = {}
    for t in tables:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        flat = _all_reduce(flat, dist.ReduceOp.SUM, group)
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


def qcmetrics_tables(qc, device, max_length: int = None) -> List[torch.Tensor]:
    """device tensors over a QCMetrics object's six tables, trimmed to max_length rows (the
    job's, once the ranks have agreed on it and the tables are padded to it)"""
    from ._lib import lib
    ptrs = (ctypes.c_void_p * 8)()
    counts = (ctypes.c_uint64 * 8)()
    k = lib().sq_qcmetrics_device_tables(qc._h, ptrs, counts, 8)
    ml, ea = qc.max_length if max_length is None else max_length, qc.end_anchor_length
    want = [ml * 5, ml * 12, ea * 5, ea * 12, 101, 94]
    return [device_tensor(ptrs[i], want[i], device) for i in range(k) if ptrs[i] and want[i]]


def adaptercounter_tables(ad, device) -> List[torch.Tensor]:
    from ._lib import lib
    ptrs = (ctypes.c_void_p * 4)()
    counts = (ctypes.c_uint64 * 4)()
    k = lib().sq_adaptercounter_device_tables(ad._h, ptrs, counts, 4)
    return [device_tensor(ptrs[i], int(counts[i]), device) for i in range(k) if ptrs[i] and counts[i]]


def agree_on_shapes(qc, ad, device, group=None) -> int:
    """Pads the count tables of this rank to the shapes of the job: QCMetrics to the longest read
    any rank saw, AdapterCounter to the longest row any rank holds (its tables are
    [adapter][row] and sq_adaptercounter_reserve grows rows geometrically, so ranks with
    different batch histories differ).  Returns the job's max_length."""
    from ._lib import check, lib, synchronize
    ml = global_max(max(qc.max_length if qc is not None else 0, ad.max_length if ad is not None else 0), device, group)
    if qc is not None:
        check(lib().sq_qcmetrics_reserve(qc._h, ml))
    if ad is not None:
        check(lib().sq_adaptercounter_reserve(ad._h, ml))
        row = global_max(int(lib().sq_adaptercounter_row_length(ad._h)), device, group)
        check(lib().sq_adaptercounter_set_row_length(ad._h, row))
    synchronize()
    return ml


def merge_qcmetrics(qc, device, group=None) -> None:
    """all ranks end up with the tables of the whole job"""
    from ._lib import check, lib, synchronize
    ml = global_max(qc.max_length, device, group)
    reads = _all_reduce(torch.tensor([qc.number_of_reads], dtype=torch.int64, device=device), group=group)
    check(lib().sq_qcmetrics_set_totals(qc._h, qc.number_of_reads, ml))   # pads to ml rows
    synchronize()
    sum_tables(qcmetrics_tables(qc, device), group)
    torch.cuda.synchronize()
    check(lib().sq_qcmetrics_set_totals(qc._h, int(reads.item()), ml))


def merge_adaptercounter(ad, device, group=None) -> None:
    from ._lib import check, lib, synchronize
    ml = agree_on_shapes(None, ad, device, group)   # same row length on every rank
    seqs = _all_reduce(torch.tensor([ad.number_of_sequences], dtype=torch.int64, device=device), group=group)
    check(lib().sq_adaptercounter_set_totals(ad._h, ad.number_of_sequences, ml))
    synchronize()
    sum_tables(adaptercounter_tables(ad, device), group)
    torch.cuda.synchronize()
    check(lib().sq_adaptercounter_set_totals(ad._h, int(seqs.item()), ml))


# ---------------------------------------------------------------------------
# the order-dependent modules
# ---------------------------------------------------------------------------
def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def _wire(t: torch.Tensor, group=None) -> torch.Tensor:
    """the tensor where the backend wants it (gloo: host, nccl/RCCL: device)"""
    return t.cpu() if dist.get_backend(group) == "gloo" else t


def _all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> torch.Tensor:
    if not _active(group):
        return t
    w = _wire(t, group).contiguous()
    dist.all_reduce(w, op=op, group=group)
    return w.to(t.device)


def all_gather_ragged(t: torch.Tensor, group=None) -> torch.Tensor:
    """concatenation over ranks, in rank order, of 1-D (or [n, k]) tensors of different n"""
    if not _active(group):
        return t
    world = dist.get_world_size(group)
    w = _wire(t, group).contiguous()
    sizes = [torch.zeros(1, dtype=torch.int64, device=w.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([w.shape[0]], dtype=torch.int64, device=w.device), group=group)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    padded = torch.zeros((cap,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    padded[:w.shape[0]] = w
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)]).to(t.device)


def _broadcast_bytes(data: bytes, src: int, device, group=None) -> bytes:
    """`data` of rank `src` on every rank"""
    n = torch.tensor([len(data)], dtype=torch.int64, device=device)
    n = _wire(n, group)
    dist.broadcast(n, src=src, group=group)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    if dist.get_rank(group) == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    buf = _wire(buf, group).contiguous()
    dist.broadcast(buf, src=src, group=group)
    return buf.cpu().numpy().tobytes()


def merge_overrepresented(shards: Sequence, device, group=None) -> None:
    """OverrepresentedSequences objects in shard mode -> every one holds the job's table:
    the first max_unique_fragments distinct fragment hashes in (sampled read, staging slot)
    order over all shards, each with its occurrences in all shards (_qcmodule.c:3543-3568)."""
    from ._lib import check, lib
    L = lib()
    hs, rs = [], []
    for o in shards:
        n = check(L.sq_overrep_shard_candidates(o._h, None, None, 0))
        h = torch.zeros(max(n, 1), dtype=torch.int64, device=device)
        r = torch.zeros(max(n, 1), dtype=torch.int64, device=device)
        torch.cuda.synchronize()   # torch fills on ITS stream; the library writes on its own: the fill must not come second
        if n:
            check(L.sq_overrep_shard_candidates(o._h, h.data_ptr(), r.data_ptr(), n))
        hs.append(h[:n])
        rs.append(r[:n])
    h_all = all_gather_ragged(torch.cat(hs), group).contiguous()
    r_all = all_gather_ragged(torch.cat(rs), group).contiguous()
    torch.cuda.synchronize()
    first = shards[0]
    sel = torch.zeros(max(min(h_all.numel(), first.max_unique_fragments), 1), dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    m = check(L.sq_overrep_shard_select(first._h, h_all.data_ptr(), r_all.data_ptr(), h_all.numel(),
                                        sel.data_ptr(), sel.numel())) if h_all.numel() else 0
    sel = sel[:m].contiguous()
    counts = torch.zeros(max(m, 1), dtype=torch.int64, device=device)
    one = torch.zeros_like(counts)
    torch.cuda.synchronize()       # as above: `one` is written by the library's stream
    for o in shards:
        if m:
            check(L.sq_overrep_shard_lookup(o._h, sel.data_ptr(), m, one.data_ptr()))   # synchronises its stream
            counts += one
            torch.cuda.synchronize()   # the next lookup overwrites `one`
    sums = torch.tensor([[o.number_of_sequences, o.sampled_sequences, o.total_fragments,
                          L.sq_overrep_warning_count(o._h)] for o in shards],
                        dtype=torch.int64, device=device).sum(0)
    last = torch.tensor([max(L.sq_overrep_last_warning_record(o._h) for o in shards)],
                        dtype=torch.int64, device=device)
    counts = _all_reduce(counts, group=group)
    sums = _all_reduce(sums, group=group)
    last = _all_reduce(last, dist.ReduceOp.MAX, group)
    torch.cuda.synchronize()
    totals = (ctypes.c_uint64 * 5)(*[int(x) for x in sums.tolist()], int(last.item()) & 0xFFFFFFFFFFFFFFFF)
    for o in shards:
        check(L.sq_overrep_shard_install(o._h, sel.data_ptr(), counts.data_ptr(), m, totals))
        o._first_record = 0
        o._warned = int(sums[3].item())


def _dedup_export(L, d) -> bytes:
    from ._lib import check
    buf = (ctypes.c_uint8 * L.sq_dedup_state_bytes(d._h))()
    check(L.sq_dedup_export_state(d._h, buf, len(buf)))
    return bytes(buf)


def _shard_counts(n_local: int, device, group=None) -> List[int]:
    """how many shards every rank holds, in rank order"""
    if not _active(group):
        return [n_local]
    t = all_gather_ragged(torch.tensor([n_local], dtype=torch.int64, device=device), group)
    return [int(x) for x in t.tolist()]


def _dedup_relay(L, shards, base: int, counts: List[int], state, start: int, device, group=None):
    """The insertion tail shard after shard, from the job's shard `start` on, beginning with
    `state` (None: a fresh estimator); shards[i] is the job's shard base + i.  Returns the
    state behind the last shard."""
    from ._lib import check

    def run_local(state):
        for i, d in enumerate(shards):
            if base + i < start:
                continue
            if state is not None:
                check(L.sq_dedup_import_state(d._h, state, len(state)))
            check(L.sq_dedup_resolve(d._h))
            state = _dedup_export(L, d)
        return state

    if not _active(group):
        return run_local(state)
    rank = dist.get_rank(group)
    first = 0
    for r, n in enumerate(counts):
        if first + n > start:          # rank r holds a shard that is still to run
            mine = run_local(state) if rank == r else b""
            state = _broadcast_bytes(mine, r, device, group)
        first += n
    return state


def dedup_store_chain(records: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """records[g] = (bytes, known) of the job's shard g, the head's all known ->
    (store_in[g], store_after[g]): a shard starts from the store the one in front leaves,
    and leaves its own bytes where it wrote and that store elsewhere (_qcmodule.c:4503-4516:
    a pair writes the front of read 1 and the back of read 2 over the buffer, short reads
    leave the rest)."""
    n = len(records)
    store_in = np.zeros((n, records.shape[2]), dtype=np.uint8)
    after = np.zeros_like(store_in)
    for g in range(n):
        if g:
            store_in[g] = after[g - 1]
        known = records[g, 1].astype(bool)
        after[g] = np.where(known, records[g, 0], store_in[g])
    return store_in, after


def _dedup_gather(L, shards, base: int, counts: List[int], device, group=None):
    """sq_ends.hip "DedupEstimator across shards, by gathering": every shard counts its
    lower bound and filters its hashes at the same time, the head (rank 0, the job's first
    shard) runs one insertion tail.  Returns the job's state."""
    from ._lib import check
    total = sum(counts)
    rank = dist.get_rank(group) if _active(group) else 0
    fp_len = check(L.sq_dedup_shard_store(shards[0]._h, 0, None, None, 0))
    # 1. the stores, and what every shard starts from
    rec = np.zeros((len(shards), 2, fp_len), dtype=np.uint8)
    for i, d in enumerate(shards):
        check(L.sq_dedup_shard_store(d._h, int(base + i == 0), rec[i, 0].ctypes.data, rec[i, 1].ctypes.data, fp_len))
    rec = all_gather_ragged(torch.from_numpy(rec.reshape(len(shards), 2 * fp_len)).to(device), group)
    rec = np.ascontiguousarray(rec.cpu().numpy()).reshape(total, 2, fp_len)
    store_in, store_after = dedup_store_chain(rec)
    # 2. settle: short pairs at the shard's start, the lower bound
    lb = np.zeros(len(shards), dtype=np.int64)
    for i, d in enumerate(shards):
        g = base + i
        out = ctypes.c_uint64(0)
        si = np.ascontiguousarray(store_in[g])
        check(L.sq_dedup_shard_settle(d._h, si.ctypes.data if g else None, fp_len, ctypes.byref(out)))
        lb[i] = out.value
    lb = all_gather_ragged(torch.from_numpy(lb).to(device), group).cpu().numpy()
    filtered = np.zeros(total, dtype=np.int64)      # filtered[g] = the largest bound in front of shard g
    for g in range(1, total):
        filtered[g] = max(filtered[g - 1], int(lb[g - 1]))
    # 3. the hashes that can still matter, to the head
    parts, sizes = [], []
    for i, d in enumerate(shards):
        g = base + i
        if g == 0:
            continue
        n = check(L.sq_dedup_shard_passing(d._h, int(filtered[g]), None, 0))
        h = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            check(L.sq_dedup_shard_passing(d._h, int(filtered[g]), h.ctypes.data, n))
        parts.append(h[:n])
        sizes.append(n)
    mine = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    sizes = all_gather_ragged(torch.tensor(sizes, dtype=torch.int64, device=device), group).cpu().numpy()
    hashes = all_gather_ragged(torch.from_numpy(mine.view(np.int64)).to(device), group)
    # 4. the head's tail: its own shard, then the others' survivors in shard order
    stop = total
    state = b""
    failure = None
    if rank == 0:
        # the head works alone while the other ranks wait in the broadcast below: whatever it raises is caught, the
        # others are told (stop = -1) and every rank raises -- a job that fails, not one that hangs
        try:
            hashes = np.ascontiguousarray(hashes.cpu().numpy()).view(np.uint64)
            head = shards[0]
            check(L.sq_dedup_resolve(head._h))
            at = 0
            for g in range(1, total):
                n = int(sizes[g - 1])
                part = np.ascontiguousarray(hashes[at:at + n])
                after = np.ascontiguousarray(store_after[g])
                rc = check(L.sq_dedup_feed_hashes(head._h, part.ctypes.data, n, int(filtered[g]), after.ctypes.data, fp_len))
                if rc:          # SQ_DEDUP_FEED_TOO_STRICT: the relay takes it from here
                    stop = g
                    break
                at += n
            state = _dedup_export(L, head)
        except Exception as e:   # noqa: BLE001 -- re-raised below, on every rank
            failure, stop, state = e, -1, b""
    if _active(group):
        t = _wire(torch.tensor([stop], dtype=torch.int64, device=device), group)
        dist.broadcast(t, src=0, group=group)
        stop = int(t.item())
        if stop >= 0:
            state = _broadcast_bytes(state, 0, device, group)
    if stop < 0:
        raise RuntimeError("merge_dedup: the head's insertion tail failed on rank 0" + (f": {failure!r}" if failure is not None else "")) from failure
    for i, d in enumerate(shards):
        if 0 < base + i < stop:
            check(L.sq_dedup_shard_drop(d._h))
    if stop < total:
        state = _dedup_relay(L, shards, base, counts, state, stop, device, group)
    return state


def merge_dedup(shards: Sequence, device=None, group=None, method: str = None) -> None:
    """DedupEstimator objects in deferred mode -> every one holds the job's estimator.
    The fingerprints were hashed in parallel and are resident.  method "relay": the
    insertion tail (_qcmodule.c:4426-4460) runs shard after shard, each continuing from the
    table, the modulo bits and the fingerprint store of the one in front (one tail and one
    broadcast of the table per shard, in sequence).  method "gather": every shard filters
    its hashes with a mask it can prove the estimator has reached by then, the head runs one
    tail over what is left (sq_ends.hip "by gathering"; falls back to the relay for the
    rest when the proof's premise fails, and for the whole merge when some rank holds no shard).  Default: $SQ_DEDUP_MERGE,
    else "relay": tests/test_gpu_shards.py runs both against one sequential run on a GPU, but with gloo ranks on ONE
    device -- until a job of more than one rank has run the gather over RCCL, the relay stays the default."""
    import os
    from ._lib import check, lib
    L = lib()
    method = method or os.environ.get("SQ_DEDUP_MERGE", "relay")
    if method not in ("relay", "gather"):
        raise ValueError(f"merge_dedup: method {method!r} (relay or gather)")
    counts = _shard_counts(len(shards), device, group)
    rank = dist.get_rank(group) if _active(group) else 0
    base = sum(counts[:rank])
    # the gather names the job's first shard through rank 0 and takes fp_len from every rank's first shard: a rank without a
    # shard would raise while the others wait in a collective.  `counts` is the same list on every rank, so every rank takes the
    # same branch: with such a rank the relay (which handles any counts) runs the whole merge
    if method == "gather" and sum(counts) > 1 and all(c > 0 for c in counts):
        state = _dedup_gather(L, shards, base, counts, device, group)
    else:
        state = _dedup_relay(L, shards, base, counts, None, 0, device, group)
    for d in shards:
        check(L.sq_dedup_set_deferred(d._h, 0))
        check(L.sq_dedup_import_state(d._h, state, len(state)))


def merge_insertsize(shards: Sequence, device, group=None) -> None:
    """InsertSizeMetrics objects in shard mode -> every one holds the job's histogram and
    adapter tables (first max_adapters distinct remainders in pair order, _qcmodule.c:5570-5611)."""
    from ._lib import check, lib
    L = lib()
    first = shards[0]
    for read2 in (0, 1):
        ks, rs = [], []
        for z in shards:
            n = check(L.sq_insertsize_shard_candidates(z._h, read2, None, None, 0))
            k = np.zeros((max(n, 1), 32), dtype=np.uint8)
            r = np.zeros(max(n, 1), dtype=np.uint64)
            if n:
                check(L.sq_insertsize_shard_candidates(z._h, read2, k.ctypes.data, r.ctypes.data, n))
            ks.append(k[:n])
            rs.append(r[:n])
        k_all = all_gather_ragged(torch.from_numpy(np.concatenate(ks)).to(device), group).cpu().numpy()
        r_all = all_gather_ragged(torch.from_numpy(np.concatenate(rs).view(np.int64)).to(device), group)
        k_all = np.ascontiguousarray(k_all)
        r_all = np.ascontiguousarray(r_all.cpu().numpy()).view(np.uint64)
        cap = max(len(r_all), 1)
        sk = np.zeros((cap, 32), dtype=np.uint8)
        sr = np.zeros(cap, dtype=np.uint64)
        m = check(L.sq_insertsize_shard_select(first._h, k_all.ctypes.data, r_all.ctypes.data, len(r_all),
                                               sk.ctypes.data, sr.ctypes.data, cap))
        counts = np.zeros(max(m, 1), dtype=np.uint64)
        one = np.zeros_like(counts)
        events = 0
        for z in shards:
            if m:
                check(L.sq_insertsize_shard_lookup(z._h, read2, sk.ctypes.data, m, one.ctypes.data))
                counts += one
            events += z.number_of_adapters_read2 if read2 else z.number_of_adapters_read1
        t = torch.from_numpy(np.concatenate([counts.view(np.int64), [events]])).to(device)
        t = _all_reduce(t, group=group).cpu().numpy()
        counts, events = np.ascontiguousarray(t[:-1]).view(np.uint64), int(t[-1])
        for z in shards:
            check(L.sq_insertsize_shard_install(z._h, read2, sk.ctypes.data, sr.ctypes.data, counts.ctypes.data,
                                                m, events))
    hists = [np.frombuffer(z.insert_sizes().tobytes(), dtype=np.uint64) for z in shards]
    length = max(len(h) for h in hists)
    if _active(group):
        length = global_max(length, device if dist.get_backend(group) != "gloo" else None, group)
    hist = np.zeros(length + 1, dtype=np.int64)
    for h in hists:
        hist[:len(h)] += h.view(np.int64)
    hist[-1] = sum(z.total_reads for z in shards)
    hist = _all_reduce(torch.from_numpy(hist).to(device), group=group).cpu().numpy()
    sizes = np.ascontiguousarray(hist[:-1]).view(np.uint64)
    for z in shards:
        check(L.sq_insertsize_shard_set_totals(z._h, int(hist[-1]), sizes.ctypes.data, len(sizes)))


def merge_pertile(shards: Sequence, first_records: Sequence[int], device, group=None) -> None:
    """PerTileQuality objects, shard i holding the job's records from first_records[i] on ->
    every one holds the job's tables.  The module stops for good at the job's first header
    without a tile id (_qcmodule.c:3126,3137-3148): shards behind it contribute nothing.
    The f64 sums are added in a different order than in one sequential run (1e-6)."""
    from ._lib import check, lib
    L = lib()
    INF = (1 << 62)
    bads = []
    for p, f in zip(shards, first_records):
        p.flush()
        b = L.sq_pertile_first_unparsable(p._h)
        bads.append(f + b if b >= 0 else INF)
    bad = int(_all_reduce(torch.tensor([min(bads)], dtype=torch.int64, device=device), dist.ReduceOp.MIN,
                          group).item())
    live = [(p, f) for p, f in zip(shards, first_records) if f <= bad]
    tables = [p.get_tile_counts() for p, _ in live]
    tiles = sorted({t for tab in tables for t, _, _ in tab})
    tiles = all_gather_ragged(torch.tensor(tiles, dtype=torch.int64, device=device), group)
    tiles = sorted(set(tiles.tolist()))
    ml = max([p.max_length for p, _ in live], default=0)
    if _active(group):
        ml = global_max(ml, device if dist.get_backend(group) != "gloo" else None, group)
    index = {t: i for i, t in enumerate(tiles)}
    errors = np.zeros((len(tiles), ml), dtype=np.float64)
    lengths = np.zeros((len(tiles), ml), dtype=np.int64)
    for tab in tables:
        for t, err, cum in tab:
            row = index[t]
            errors[row, :len(err)] += np.asarray(err, dtype=np.float64)
            c = np.asarray(list(cum) + [0], dtype=np.int64)
            lengths[row, :len(cum)] += c[:-1] - c[1:]  # undo the reverse cumulation (:3336-3347)
    reads = sum(p.number_of_reads for p, _ in live)
    reason = b""
    for (p, f), b in zip(zip(shards, first_records), bads):
        if b == bad and bad != INF:
            reason = (p.skipped_reason or "").encode()
    rbuf = np.zeros(8192, dtype=np.int64)
    rbuf[:len(reason)] = np.frombuffer(reason, dtype=np.uint8)[:8192]
    errors_t = _all_reduce(torch.from_numpy(errors).to(device), group=group).cpu().numpy()
    ints = np.concatenate([lengths.reshape(-1), [reads], rbuf])
    ints = _all_reduce(torch.from_numpy(ints).to(device), group=group).cpu().numpy()
    lengths = np.ascontiguousarray(ints[:lengths.size]).view(np.uint64)
    reads = int(ints[lengths.size])
    reason = bytes(ints[lengths.size + 1:].astype(np.uint8)).rstrip(b"\0")
    errors_t = np.ascontiguousarray(errors_t)
    ids = np.asarray(tiles, dtype=np.int64)
    for p in shards:
        check(L.sq_pertile_install(p._h, ids.ctypes.data, len(ids), errors_t.ctypes.data, lengths.ctypes.data,
                                   ml, reads, reason if bad != INF else None))
