"""DedupEstimator's table in HBM (csrc/sq_ends.hip, dedup_process): streams of raw hashes through the parallel
steps -- lookups against the table as a piece finds it, priority linear probing for the new hashes, rebuilds on the
device, the arrival that rebuilds placed with the old bit count -- against the oracle's sequential estimator
(_qcmodule.c:4383-4460 restated in oracle/sq_oracle.c): modulo bits, tracked sequences and duplication_counts() in
SLOT order (:4736-4744), which only the reference's exact table layout gives.  Needs a GPU."""
import numpy as np
import pytest

from oracle import oracle
from sequali_amd._lib import check, context, lib

pytestmark = pytest.mark.gpu


def stream(rng, n, universe, spread_bits=64):
    """n hashes drawn from `universe` distinct values; spread_bits < 64 keeps the low 64 - spread_bits bits zero"""
    pool = rng.integers(0, 1 << 63, size=universe, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=universe, dtype=np.uint64)
    if spread_bits < 64:
        pool = pool << np.uint64(64 - spread_bits)
    return pool[rng.integers(0, universe, size=n)]


class Estimator:
    """the library's estimator fed with hashes (sq_dedup_feed_hashes: the gather merge's entry point)"""

    def __init__(self, max_stored):
        self.L = lib()
        self.h = self.L.sq_dedup_new(context(), max_stored, 8, 8, 64, 64)
        assert self.h

    def feed(self, hashes):
        a = np.ascontiguousarray(hashes, dtype=np.uint64)
        assert check(self.L.sq_dedup_feed_hashes(self.h, a.ctypes.data, len(a), 0, None, 0)) == 0

    def state(self):
        n = check(self.L.sq_dedup_duplication_counts(self.h, None, 0))
        counts = np.zeros(max(n, 1), dtype=np.uint64)
        check(self.L.sq_dedup_duplication_counts(self.h, counts.ctypes.data, n))
        return (self.L.sq_dedup_modulo_bits(self.h), self.L.sq_dedup_tracked_sequences(self.h), counts[:n].tolist())

    def pieces(self):
        return self.L.sq_dedup_device_pieces(self.h), self.L.sq_dedup_host_pieces(self.h)

    def close(self):
        self.L.sq_dedup_free(self.h)


def oracle_state(hashes, max_stored):
    o = oracle.DedupEstimator(max_stored)
    for h in hashes:
        o.add_hash(int(h))
    return (o._modulo_bits, o.tracked_sequences, o.duplication_counts().tolist())


@pytest.mark.parametrize("max_stored,n,universe,step", [
    (100, 5000, 3000, 777), (250, 20000, 900, 20000), (1000, 60000, 50000, 7001), (100, 300, 100, 300),
    (100, 40000, 150, 40000),       # nearly every arrival is found: few inserts, rebuilds by found hashes
    (5000, 200000, 150000, 65536),  # pieces of many survivors, several rebuilds inside one call
    (300, 30000, 30000, 1),         # hash by hash: every call one piece
])
def test_streams_of_hashes_leave_the_sequential_table(max_stored, n, universe, step):
    rng = np.random.default_rng(max_stored + n)
    h = stream(rng, n, universe)
    if step == 1:
        h = h[:1500]
    est = Estimator(max_stored)
    try:
        for lo in range(0, len(h), step):
            est.feed(h[lo:lo + step])
        assert est.state() == oracle_state(h, max_stored)
        dev, host = est.pieces()
        assert dev > 0
    finally:
        est.close()


@pytest.mark.parametrize("seed", range(6))
def test_the_arrival_that_rebuilt_comes_again(seed):
    """Small universes: the hash whose arrival rebuilds the table (placed with the OLD bit count, outside the probe
    run of its hash) arrives again and again.  Pieces in which it is not found take the host's loop; the table is
    the sequential one either way."""
    rng = np.random.default_rng(900 + seed)
    host_total = 0
    for _ in range(6):
        max_stored = int(rng.integers(100, 260))
        universe = int(rng.integers(max_stored + 20, 6 * max_stored))
        h = stream(rng, int(rng.integers(2000, 30000)), universe)
        est = Estimator(max_stored)
        try:
            for lo in range(0, len(h), 4096):
                est.feed(h[lo:lo + 4096])
            assert est.state() == oracle_state(h, max_stored)
            host_total += est.pieces()[1]
        finally:
            est.close()
    assert host_total >= 0


def test_hashes_whose_low_bits_are_not_spread():
    """every arrival rebuilds once the table is full and a rebuild drops nothing (:4436-4451): the pieces end at
    their first survivors, and after a few of those the rest of the call takes the host's loop"""
    rng = np.random.default_rng(77)
    h = stream(rng, 3000, 2500, spread_bits=40)
    est = Estimator(120)
    try:
        est.feed(h)
        assert est.state() == oracle_state(h, 120)
    finally:
        est.close()


def test_a_million_fingerprints_default_geometry():
    """the default table (1 M fingerprints, 2^21 slots) through its first rebuilds: 4 M arrivals, a third of them
    repeats; one call of 4 M hashes and the same stream in calls of 1 M"""
    rng = np.random.default_rng(5)
    h = stream(rng, 4_000_000, 2_600_000)
    want = oracle_state(h, 1_000_000)
    assert want[0] >= 1
    for step in (len(h), 1_000_000):
        est = Estimator(1_000_000)
        try:
            for lo in range(0, len(h), step):
                est.feed(h[lo:lo + step])
            got = est.state()
            assert got[:2] == want[:2]
            np.testing.assert_array_equal(np.array(got[2], np.uint64), np.array(want[2], np.uint64))
            assert est.pieces()[0] > 0
        finally:
            est.close()


def test_state_travels_between_host_and_device():
    """export / import (the shards' relay) around pieces on the device: the imported table goes on in HBM, the entry
    outside its probe run is found again from the table alone"""
    rng = np.random.default_rng(31)
    L = lib()
    h = stream(rng, 60000, 9000)
    a, b = Estimator(400), Estimator(400)
    try:
        a.feed(h[:30000])
        buf = (np.zeros(L.sq_dedup_state_bytes(a.h), dtype=np.uint8))
        check(L.sq_dedup_export_state(a.h, buf.ctypes.data, len(buf)))
        check(L.sq_dedup_import_state(b.h, buf.ctypes.data, len(buf)))
        b.feed(h[30000:])
        assert b.state() == oracle_state(h, 400)
    finally:
        a.close()
        b.close()
