"""DedupEstimator across shards by gathering (sq_ends.hip "by gathering", dist._dedup_gather):
the host halves, which need no device -- the lower bound from a histogram of trailing zero
bits, the chain of fingerprint stores, and the head's insertion tail over filtered hashes --
against the oracle's sequential estimator (_qcmodule.c:4426-4460 restated in oracle/).

The device halves (the sort + histogram that counts the bound, the ordered filter) are
covered by tests/test_gpu_shards.py with method="gather"."""
import ctypes

import numpy as np
import pytest

from oracle import oracle
from sequali_amd import dist
from sequali_amd._lib import check, lib


def ctz(h: int) -> int:
    return 64 if h == 0 else (h & -h).bit_length() - 1


def lower_bound(hashes, max_stored: int) -> int:
    """the definition: least b such that at most max_stored DISTINCT hashes have b trailing zero bits"""
    distinct = set(int(h) for h in hashes)
    b = 0
    while b < 63 and sum(1 for h in distinct if h & ((1 << b) - 1) == 0) > max_stored:
        b += 1
    return b


def lower_bound_c(hashes, max_stored: int) -> int:
    hist = np.zeros(65, dtype=np.uint64)
    for h in set(int(h) for h in hashes):
        hist[ctz(h)] += 1
    return int(lib().sq_dedup_lower_bound_of(hist.ctypes.data, max_stored))


def stream(rng, n, universe, spread_bits=64):
    """n hashes drawn from `universe` distinct values; spread_bits < 64 keeps the low
    64 - spread_bits bits zero (hashes whose low bits are not spread)"""
    pool = rng.integers(0, 1 << 63, size=universe, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=universe, dtype=np.uint64)
    if spread_bits < 64:
        pool = pool << np.uint64(64 - spread_bits)
    return pool[rng.integers(0, universe, size=n)]


class Head:
    """the library's estimator without a device: sq_dedup_new touches none, and the feed is the host's tail"""

    def __init__(self, max_stored):
        self.L = lib()
        self.h = self.L.sq_dedup_new(None, max_stored, 8, 8, 64, 64)
        assert self.h

    def feed(self, hashes, filtered_bits=0, store_after=None):
        a = np.ascontiguousarray(hashes, dtype=np.uint64)
        s = None if store_after is None else np.ascontiguousarray(store_after, dtype=np.uint8)
        return check(self.L.sq_dedup_feed_hashes(self.h, a.ctypes.data, len(a), filtered_bits,
                                                 s.ctypes.data if s is not None else None, 0 if s is None else len(s)))

    def state(self):
        n = check(self.L.sq_dedup_duplication_counts(self.h, None, 0))
        counts = np.zeros(max(n, 1), dtype=np.uint64)
        check(self.L.sq_dedup_duplication_counts(self.h, counts.ctypes.data, n))
        return (self.L.sq_dedup_modulo_bits(self.h), self.L.sq_dedup_tracked_sequences(self.h),
                self.L.sq_dedup_hash_table_size(self.h), counts[:n].tolist())

    def close(self):
        self.L.sq_dedup_free(self.h)


def oracle_state(hashes, max_stored):
    o = oracle.DedupEstimator(max_stored)
    for h in hashes:
        o.add_hash(int(h))
    return (o._modulo_bits, o.tracked_sequences, o._hash_table_size, o.duplication_counts().tolist())


@pytest.mark.parametrize("seed", range(6))
def test_lower_bound_from_histogram(seed):
    rng = np.random.default_rng(seed)
    for _ in range(20):
        max_stored = int(rng.integers(100, 400))
        h = stream(rng, int(rng.integers(0, 6000)), int(rng.integers(1, 5000)), int(rng.choice([64, 64, 60, 40])))
        assert lower_bound_c(h, max_stored) == lower_bound(h, max_stored)
    assert lower_bound_c([0] * 10, 100) == 0
    assert lower_bound_c([], 100) == 0


def test_feed_is_the_sequential_tail():
    """sq_dedup_feed_hashes, unfiltered, is DedupEstimator_add_fingerprint's tail: bits, slots, counts"""
    rng = np.random.default_rng(11)
    for max_stored, n, universe in ((100, 5000, 3000), (250, 20000, 900), (1000, 60000, 50000), (100, 300, 100)):
        h = stream(rng, n, universe)
        head = Head(max_stored)
        try:
            for lo in range(0, n, 777):      # in several calls, like several shards
                assert head.feed(h[lo:lo + 777]) == 0
            assert head.state() == oracle_state(h, max_stored)
        finally:
            head.close()


def gather_merge(shards, max_stored, head):
    """dist._dedup_gather's steps 2-4 on host arrays; returns the shard the feed refused (or len(shards))"""
    lbs = [lower_bound_c(s, max_stored) for s in shards]
    assert head.feed(shards[0]) == 0
    filtered = 0
    for g in range(1, len(shards)):
        filtered = max(filtered, lbs[g - 1])
        s = shards[g]
        passing = s[(s & np.uint64((1 << filtered) - 1)) == 0]
        if head.feed(passing, filtered) == 1:
            return g
    return len(shards)


@pytest.mark.parametrize("seed", range(8))
def test_gather_is_the_sequential_estimator(seed):
    rng = np.random.default_rng(100 + seed)
    for _ in range(12):
        max_stored = int(rng.integers(100, 300))
        n_shards = int(rng.integers(2, 7))
        universe = int(rng.integers(50, 20000))
        shards = [stream(rng, int(rng.integers(0, 4000)), universe) for _ in range(n_shards)]
        head = Head(max_stored)
        try:
            assert gather_merge(shards, max_stored, head) == n_shards   # spread hashes: the premise holds
            assert head.state() == oracle_state(np.concatenate(shards), max_stored)
        finally:
            head.close()


def test_gather_filters_most_of_a_late_shard():
    """what the merge is for: the head sees about max_stored << 1 hashes of a later shard, not the shard"""
    rng = np.random.default_rng(5)
    max_stored = 200
    shards = [stream(rng, 50000, 40000) for _ in range(4)]
    lbs = [lower_bound_c(s, max_stored) for s in shards]
    assert min(lbs) >= 6
    passing = int(((shards[1] & np.uint64((1 << lbs[0]) - 1)) == 0).sum())
    assert passing < len(shards[1]) // 32
    head = Head(max_stored)
    try:
        assert gather_merge(shards, max_stored, head) == 4
        assert head.state() == oracle_state(np.concatenate(shards), max_stored)
    finally:
        head.close()


def test_feed_refuses_a_filter_the_estimator_has_not_reached():
    """Hashes with no spread in the low bits: every rebuild keeps everything, the table overfills (:4436-4451 rebuilds
    once per arriving hash) and the bits lag behind the bound.  The feed says so and does nothing; feeding the rest
    unfiltered -- what the relay does -- is exact."""
    rng = np.random.default_rng(9)
    max_stored = 100
    first = np.unique(stream(rng, 200, 200, spread_bits=24))[:115]     # 115 distinct: 15 rebuilds that drop nothing
    rng.shuffle(first)
    shards = [first, stream(rng, 400, 300, spread_bits=24), stream(rng, 500, 5000)]
    assert len(first) == 115 and lower_bound_c(shards[0], max_stored) >= 40
    head = Head(max_stored)
    try:
        before = None
        stop = None
        lbs = [lower_bound_c(s, max_stored) for s in shards]
        assert head.feed(shards[0]) == 0
        if head.state()[0] < lbs[0]:
            before = head.state()
            s = shards[1]
            assert head.feed(s[(s & np.uint64((1 << lbs[0]) - 1)) == 0], lbs[0]) == 1
            assert head.state() == before
            stop = 1
        assert stop == 1, "the case did not provoke the refusal"
        for s in shards[stop:]:
            assert head.feed(s) == 0
        assert head.state() == oracle_state(np.concatenate(shards), max_stored)
    finally:
        head.close()


def test_store_chain():
    """a shard starts from the store the one in front leaves; bytes it did not write pass through (:4503-4516)"""
    fp = 6
    rec = np.zeros((4, 2, fp), dtype=np.uint8)
    rec[0, 0] = [1, 2, 3, 4, 5, 6]; rec[0, 1] = 1                     # the head: all known
    rec[1, 0] = [9, 9, 0, 0, 9, 0]; rec[1, 1] = [1, 1, 0, 0, 1, 0]     # wrote three bytes
    rec[2, 1] = 0                                                     # wrote nothing (single-end, or empty)
    rec[3, 0] = [7, 7, 7, 7, 7, 7]; rec[3, 1] = 1
    store_in, after = dist.dedup_store_chain(rec)
    assert after[0].tolist() == [1, 2, 3, 4, 5, 6]
    assert store_in[1].tolist() == [1, 2, 3, 4, 5, 6] and after[1].tolist() == [9, 9, 3, 4, 9, 6]
    assert store_in[2].tolist() == after[1].tolist() == after[2].tolist()
    assert store_in[3].tolist() == after[2].tolist() and after[3].tolist() == [7] * 6


def test_feed_sets_the_store_and_checks_its_length():
    head = Head(100)
    try:
        assert head.feed([], 0, np.arange(16, dtype=np.uint8)) == 0
        with pytest.raises(ValueError):
            head.feed([], 0, np.arange(5, dtype=np.uint8))
        buf = (ctypes.c_uint8 * lib().sq_dedup_state_bytes(head.h))()
        check(lib().sq_dedup_export_state(head.h, buf, len(buf)))
        assert bytes(buf)[40:56] == bytes(range(16))
    finally:
        head.close()


# ---- the collective flow of dist.merge_dedup under gloo, without a device -----------------------
class _StreamLib:
    """libsqgpu.so with the five entry points that read the resident stream (HBM) replaced by the
    same steps on a host array; the estimator, its tail, export and import are the library's.
    What this leaves to the GPU tests: the device sort + histogram, the ordered filter."""

    def __init__(self):
        self._L = lib()
        self.streams = {}      # handle -> [hashes (np.uint64), store bytes, known]

    def __getattr__(self, name):
        return getattr(self._L, name)

    def new(self, max_stored, hashes, store, known):
        h = self._L.sq_dedup_new(None, max_stored, 4, 4, 0, 0)
        check(self._L.sq_dedup_set_deferred(h, 1))
        self.streams[h] = [np.ascontiguousarray(hashes, dtype=np.uint64), np.array(store, dtype=np.uint8), np.array(known, dtype=np.uint8)]
        return h

    @staticmethod
    def _at(ptr, n):
        return np.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=np.uint8)

    def sq_dedup_shard_store(self, h, head, bytes_ptr, known_ptr, cap):
        _, store, known = self.streams[h]
        if bytes_ptr and known_ptr:
            self._at(bytes_ptr, len(store))[:] = store
            self._at(known_ptr, len(store))[:] = 1 if head else known
        return len(store)

    def sq_dedup_shard_settle(self, h, store_in, n, lower_bound):
        lower_bound._obj.value = lower_bound_c(self.streams[h][0], 120)
        return 0

    def sq_dedup_shard_passing(self, h, bits, out, cap):
        s = self.streams[h][0]
        keep = s[(s & np.uint64((1 << bits) - 1)) == 0]
        if out:
            np.frombuffer((ctypes.c_uint64 * len(keep)).from_address(out), dtype=np.uint64)[:] = keep
        return len(keep)

    def sq_dedup_shard_drop(self, h):
        self.streams[h][0] = np.zeros(0, dtype=np.uint64)
        return 0

    def sq_dedup_resolve(self, h):
        s = self.streams[h][0]
        rc = self._L.sq_dedup_feed_hashes(h, s.ctypes.data, len(s), 0, None, 0)
        self.streams[h][0] = np.zeros(0, dtype=np.uint64)
        return rc


class _Shard:
    def __init__(self, h):
        self._h = h


def _job(case):
    """the job's shards: (hashes, store bytes, known) per shard, seeded; shards per rank"""
    rng = np.random.default_rng(7)
    per_rank = {"plain": [1, 1], "three": [1, 2, 1], "lagging": [1, 1, 1]}[case]
    shards = []
    for g in range(sum(per_rank)):
        if case == "lagging" and g == 0:
            h = np.unique(stream(rng, 300, 300, spread_bits=24))[:130]   # more distinct than fit, no spread: see above
        else:
            h = stream(rng, int(rng.integers(500, 5000)), 6000, 24 if case == "lagging" and g == 1 else 64)
        known = rng.integers(0, 2, size=8).astype(np.uint8) if g else np.ones(8, dtype=np.uint8)
        shards.append((h, rng.integers(0, 255, size=8).astype(np.uint8), known))
    return per_rank, shards


def _gloo_rank(rank, world, port, case, method, out_dir):
    import os
    import torch.distributed as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    per_rank, job = _job(case)
    base = sum(per_rank[:rank])
    L = _StreamLib()
    mine = [_Shard(L.new(120, *job[base + i])) for i in range(per_rank[rank])]
    counts = dist._shard_counts(len(mine), None)
    assert counts == per_rank
    if method == "gather":
        state = dist._dedup_gather(L, mine, base, counts, None)
    else:
        state = dist._dedup_relay(L, mine, base, counts, None, 0, None)
    with open(os.path.join(out_dir, f"state{rank}.bin"), "wb") as f:
        f.write(state)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("case,method", [("plain", "gather"), ("three", "gather"), ("lagging", "gather"), ("three", "relay")])
def test_merge_flow_over_gloo(tmp_path, case, method):
    """two / three processes, one of them with two shards; "lagging": the head's bits are behind the bound
    of its shard, the feed refuses shard 1 and the relay finishes from there -- the state is the oracle's
    sequential one on every rank (bits, tracked, slots in order; the store by the chain)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    per_rank, job = _job(case)
    mp.spawn(_gloo_rank, args=(len(per_rank), port, case, method, str(tmp_path)), nprocs=len(per_rank), join=True)
    want = oracle_state(np.concatenate([j[0] for j in job]), 120)
    rec = np.stack([np.stack([j[1], j[2]]) for j in job])
    store = dist.dedup_store_chain(rec)[1][-1]
    states = [open(tmp_path / f"state{r}.bin", "rb").read() for r in range(len(per_rank))]
    assert all(s == states[0] for s in states)
    head = np.frombuffer(states[0][:40], dtype=np.uint64)
    bits, stored, table_size, fp_len = (int(x) for x in head[1:])
    assert (bits, stored, table_size, fp_len) == (want[0], want[1], want[2], 8)
    counts = np.frombuffer(states[0][48 + table_size * 8:], dtype=np.uint32)
    assert counts[counts != 0].tolist() == want[3]
    if method == "gather" and case != "lagging":
        assert states[0][40:48] == store.tobytes()


# ---- the tail's walk over pairs shorter than the fingerprint, as a model --------------------------------------
def _pair_store_bytes(s1, s2, fl0, bl0, fo0, bo0):
    """what a pair writes into the fingerprint store (_qcmodule.c:4503-4514; pair_store_bytes in sq_ends.hip)"""
    fl = min(fl0, len(s1)); fo = min(fo0, len(s1) - fl)
    bl = min(bl0, len(s2)); bo = min(bo0, len(s2) - bl)
    return s1[fo:fo + fl] + s2[bo:bo + bl], len(s1) + len(s2)


def _tail_model(batches, max_stored, fl0, bl0, fo0, bo0):
    """dedup_run / dedup_tail of sq_ends.hip restated: long pairs are hashed on their own (the device's k_dedup_hash), a
    short pair's fingerprint is the store with its bytes laid over -- the store CARRIED from short pair to short pair of
    a batch (run_store, run_prev), taken from the pair in front when that one was long, and handed from batch to batch"""
    fp = fl0 + bl0
    est = oracle.DedupEstimator(max_stored, front_sequence_length=fl0, back_sequence_length=bl0,
                                front_sequence_offset=fo0, back_sequence_offset=bo0)
    store = bytearray(fp)
    for s1, s2 in batches:
        run_store, run_prev = bytearray(store), None
        for r in range(len(s1)):
            w, total = _pair_store_bytes(s1[r], s2[r], fl0, bl0, fo0, bo0)
            if len(w) == fp:
                h = oracle.murmur3_x64_64(bytes(w), total >> 6)
            else:
                if r != 0 and run_prev != r - 1:
                    run_store = bytearray(_pair_store_bytes(s1[r - 1], s2[r - 1], fl0, bl0, fo0, bo0)[0])
                run_store[:len(w)] = w
                run_prev = r
                h = oracle.murmur3_x64_64(bytes(run_store), total >> 6)
            est.add_hash(h)
        if len(s1):
            store = bytearray(run_store) if run_prev == len(s1) - 1 else bytearray(_pair_store_bytes(s1[-1], s2[-1], fl0, bl0, fo0, bo0)[0])
    return est


@pytest.mark.parametrize("seed", range(6))
def test_the_walk_over_short_pairs_is_the_sequential_store(seed):
    """the ALGORITHM of the library's tail for pairs shorter than the fingerprint (round 4: the store is carried instead of
    rebuilt by walking back from every short pair) against the oracle's pair by pair -- several batches, four geometries,
    batches of nothing but short pairs.  The library's own run of it needs a GPU
    (tests/test_gpu_vs_oracle.py::test_dedup_batches_of_nothing_but_short_pairs)."""
    rng = np.random.default_rng(300 + seed)
    for _ in range(25):
        max_len = int(rng.choice([3, 7, 9, 12, 20]))
        geom = (int(rng.choice([8, 3])), int(rng.choice([8, 5])), int(rng.choice([0, 4])), int(rng.choice([0, 2])))
        ref = oracle.DedupEstimator(100, front_sequence_length=geom[0], back_sequence_length=geom[1],
                                    front_sequence_offset=geom[2], back_sequence_offset=geom[3])
        batches = []
        for _ in range(int(rng.integers(1, 4))):
            n = int(rng.choice([1, 2, 50, 400]))
            s1 = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, max_len + 1))).tobytes() for _ in range(n)]
            s2 = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, max_len + 1))).tobytes() for _ in range(n)]
            batches.append((s1, s2))
            for a, b in zip(s1, s2):
                ref.add_sequence_pair(a.decode(), b.decode())
        got = _tail_model(batches, 100, *geom)
        assert (got._modulo_bits, got.tracked_sequences, got.duplication_counts().tolist()) == \
            (ref._modulo_bits, ref.tracked_sequences, ref.duplication_counts().tolist())
