"""The passes of config 3 that carry more than one module (csrc/sq_pair.hip): PerTileQuality inside QCMetrics' pass
(k_span<PT>: illumina_header_to_tile_id, _qcmodule.c:3088-3121, on the header bytes the pass fetches anyway;
PerTileQuality_add_meta :3123-3222 summed per run of reads of one tile and folded into the tables afterwards), against
the oracle and against the passes of round 2 (SQ_PT_FUSED=0).  Needs a GPU."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import with_env
from tests.test_gpu_vs_oracle import _route_of, compare_qc, u64

pytestmark = pytest.mark.gpu

RIDE = {"SQ_PT_FUSED": "1"}    # the pass under test (not the default yet: csrc/sq_common.h, SqKnobs::pt_fused)

LETTERS = np.frombuffer(b"ACGT", np.uint8)


def _batch(rng, tiles_of_reads, U, name_of=None, qual_hi=94):
    names, seqs, quals = [], [], []
    for i, t in enumerate(tiles_of_reads):
        names.append(name_of(i, t) if name_of else f"M0:7:FCX:{1 + i % 4}:{t}:{1000 + i}:{U} 1:N:0:ACGT")
        seqs.append(rng.choice(LETTERS, size=U).tobytes().decode())
        quals.append((rng.integers(0, qual_hi, size=U) + 33).astype(np.uint8).tobytes().decode())
    return oracle.make_batch(names, seqs, quals)


def _compare_pertile(g, r):
    assert g.number_of_reads == r.number_of_reads
    assert g.max_length == r.max_length
    gt, rt = g.get_tile_counts(), r.get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (t, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6, atol=0, err_msg=f"tile {t}")
        np.testing.assert_array_equal(u64(c), cr, err_msg=f"tile {t}")


def _runs(rng, n, lengths, tiles):
    """tile of every read: runs whose lengths are drawn from `lengths`, tiles from `tiles`"""
    out = []
    while len(out) < n:
        out += [int(tiles[int(rng.integers(0, len(tiles)))])] * int(lengths[int(rng.integers(0, len(lengths)))])
    return out[:n]


@pytest.mark.parametrize("U,n", [(150, 16 * 700), (150, 16 * 700 + 11), (200, 16 * 300 + 5), (31, 16 * 400 + 1), (256, 16 * 260), (97, 4096)])
def test_pertile_rides_in_the_qcmetrics_pass(U, n):
    """reads that come tile by tile, in runs of 1 to a few thousand reads (tile changes at every offset inside a span of
    16; runs that end with the batch; runs that go on in the next batch), two batches one after the other; the records
    behind the last full span (k_pt_tail).  The route is asserted: nothing falls back."""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(U * 1000 + n)
    tiles = [1101, 1102, 1203, 2101, 2224, 7, 0, 99999999, 123456789012]   # 0, 8 and 12 digits: the 64-bit paths of the parse
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    gq, gp = QCMetrics(), PerTileQuality()
    f = FusedPass(gq, None, gp)
    nw = (U + 31) // 32
    for part in range(2):
        tl = _runs(rng, n, [1, 2, 5, 15, 16, 17, 33, 250, 3000], tiles)
        buf, metas = _batch(rng, tl, U)
        rq.add(buf, metas)
        rp.add(buf, metas)
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        r = _route_of(lambda: with_env(RIDE, lambda: (f.add_record_array(arr), gq.flush(), gp.flush())))
        assert r.split("+")[0] == f"k_span<{nw},QCPT,uniform,both>" and "k_pt_fold" in r and "k_ptspan" not in r and "k_tile" not in r, r
        compare_qc(rq, gq, metas, arr)
        _compare_pertile(gp, rp)


def test_pertile_ride_headers_of_every_shape():
    """headers longer than the 64 bytes the pass looks at (the tile field in front of and behind byte 64), a tile field
    of 18 digits, read-2 style comments, names of different lengths inside one span"""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(5)
    U, n = 100, 16 * 300

    def name_of(i, t):
        k = i % 6
        if k == 0:
            return f"A:1:F:2:{t}:5:6"
        if k == 1:
            return f"INSTRUMENT-WITH-A-LONG-NAME-0001:123456:FLOWCELLXX:4:{t}:12345:67890 2:N:0:ACGTACGT+TTGCAAGC extra words here"
        if k == 2:
            return "X" * 61 + f":1:F:2:{t}:5:6"        # the fourth colon behind byte 64
        if k == 3:
            return "Y" * 50 + f":1:F:2:{t}:5:6 tail"   # the tile field straddles byte 64
        if k == 4:
            return f"::::{t}:"
        return f"M:1:F:{i % 4}:{t}:{i}:{U} 1:N:0:X"

    tl = _runs(rng, n, [16, 32, 48, 64, 160], [5, 1101, 123456789012345678, 22])
    buf, metas = _batch(rng, tl, U, name_of)
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    rq.add(buf, metas)
    rp.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, gp = QCMetrics(), PerTileQuality()
    r = _route_of(lambda: with_env(RIDE, lambda: (FusedPass(gq, None, gp).add_record_array(arr), gq.flush(), gp.flush())))
    assert r.split("+")[0] == "k_span<4,QCPT,uniform,both>" and "k_pt_fold" in r, r
    compare_qc(rq, gq, metas, arr)
    _compare_pertile(gp, rp)
    assert gp.skipped_reason is None


@pytest.mark.parametrize("bad_at", [0, 5000, 16 * 400 - 1, 16 * 400 + 3])
def test_pertile_ride_meets_a_header_that_does_not_parse(bad_at):
    """the reference stops PerTileQuality at the first header that does not parse (:3137-3148) and keeps what it has
    counted: the runs the pass staged are dropped and the older route counts the batch up to that record"""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(bad_at)
    U, n = 150, 16 * 400 + 7

    def name_of(i, t):
        return "no colons here" if i == bad_at else f"M:1:F:{i % 4}:{t}:{i}:{U} 1:N:0:X"

    tl = _runs(rng, n, [100, 1000], [1101, 1102, 1103])
    buf, metas = _batch(rng, tl, U, name_of)
    buf2, metas2 = _batch(rng, tl, U)
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    gq, gp = QCMetrics(), PerTileQuality()
    f = FusedPass(gq, None, gp)
    for b, m in ((buf2, metas2), (buf, metas), (buf2, metas2)):   # a clean batch, the bad one, one more (ignored: :3126)
        rq.add(b, m)
        rp.add(b, m)
        arr = FastqRecordArrayView._from_buffer(b, m.copy())
        with_env(RIDE, lambda: (f.add_record_array(arr), gq.flush()))
        compare_qc(rq, gq, m, arr)
    _compare_pertile(gp, rp)
    assert rp.skipped and gp.skipped_reason == "Can not parse header: 'no colons here'"
    assert gp.number_of_reads == n + bad_at


def test_pertile_ride_gives_way_to_reads_of_mixed_tiles():
    """reads of random tiles make a run per read: the staging area overflows, the batch is counted by k_ptspan, and the
    following batches only take their tile ids from the pass (no k_tile_parse any more)"""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(77)
    U, n = 150, 16 * 1024
    tiles = [1000 + t for t in range(40)]
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    gq, gp = QCMetrics(), PerTileQuality()
    f = FusedPass(gq, None, gp)
    routes = []
    for part in range(3):
        tl = [int(tiles[int(x)]) for x in rng.integers(0, len(tiles), size=n)]
        buf, metas = _batch(rng, tl, U)
        rq.add(buf, metas)
        rp.add(buf, metas)
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        routes.append(_route_of(lambda: with_env(RIDE, lambda: (f.add_record_array(arr), gq.flush(), gp.flush()))))
        compare_qc(rq, gq, metas, arr)
        _compare_pertile(gp, rp)
    assert routes[0].startswith("k_span<5,QCPT,uniform,both>") and "k_ptspan<5>" in routes[0] and "k_pt_fold" not in routes[0], routes[0]
    for r in routes[1:]:
        assert r == "k_span<5,QCPT,uniform,both>+k_ptspan<5>", r


@pytest.mark.parametrize("fused", ["0", "2"])
def test_pertile_ride_switched_off_or_tile_ids_only(fused):
    """SQ_PT_FUSED=0: the passes of round 2 (k_tile_parse, k_span, k_ptspan); 2: tile ids from the pass, the table by
    k_ptspan -- the cross-checks of the default"""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(int(fused))
    U, n = 150, 16 * 500 + 9
    tl = _runs(rng, n, [1, 40, 700], [1101, 1102, 2101])
    buf, metas = _batch(rng, tl, U)
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    rq.add(buf, metas)
    rp.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, gp = QCMetrics(), PerTileQuality()
    r = _route_of(lambda: with_env({"SQ_PT_FUSED": fused}, lambda: (FusedPass(gq, None, gp).add_record_array(arr), gq.flush(), gp.flush())))
    if fused == "0":
        assert r.startswith("k_span<5,QC,uniform,both>") and "k_ptspan<5>" in r, r
    else:
        assert r.startswith("k_span<5,QCPT,uniform,both>") and "k_ptspan<5>" in r and "k_pt_fold" not in r, r
    compare_qc(rq, gq, metas, arr)
    _compare_pertile(gp, rp)


def test_pertile_ride_on_device_batches_by_tile():
    """the bench's config 3 shape at a small size: device-generated reads in the order a sequencer writes (65536 of a
    tile in a row), both mates"""
    from sequali_amd import FusedPass, PerTileQuality, QCMetrics, synth
    n, first = 300_000, 65536 * 96 - 100_000     # the tile numbers wrap inside the batch
    for kind in (synth.ILLUMINA_BY_TILE, synth.ILLUMINA_R2_BY_TILE):
        dev = synth.device_array(kind, first, n)
        buf, metas = dev._batch.download()
        rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
        rq.add(buf, metas)
        rp.add(buf, metas)
        gq, gp = QCMetrics(), PerTileQuality()
        r = _route_of(lambda: with_env(RIDE, lambda: (FusedPass(gq, None, gp).add_record_array(dev), gq.flush(), gp.flush())))
        assert r == "k_span<5,QCPT,uniform,both>+k_pt_fold", r
        compare_qc(rq, gq, metas, dev)
        _compare_pertile(gp, rp)
        assert len(rp.get_tile_counts()) >= 5
