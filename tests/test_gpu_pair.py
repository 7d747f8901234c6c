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
    # (tile ids the ORACLE can hold: like the reference it indexes an array by the id, 16 bytes per id up to the largest --
    # ids of 8 and more digits are checked without it, test_pertile_ride_tile_ids_of_many_digits)
    tiles = [1101, 1102, 1203, 2101, 2224, 7, 0, 65535, 40000]
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
    """headers longer than the 64 bytes the pass looks at (the tile field in front of and behind byte 64), read-2 style
    comments, names of different lengths inside one span"""
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

    tl = _runs(rng, n, [16, 32, 48, 64, 160], [5, 1101, 31999, 22])
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


def test_pertile_ride_tile_ids_of_many_digits():
    """tile ids of 8, 9, 12 and 18 digits (the reference takes up to 18, :159-180; the pass parses up to 8 in registers and
    the rest byte by byte).  NOT against the oracle: like the reference it indexes an array by the tile id and would ask
    for terabytes (that took two GPU boxes down in round 4); the expectation is made here from the definition --
    total_errors[tile][pos] = sum of 10^(-q/10) over the tile's reads, length_counts = the reads -- within the module's
    1e-6"""
    from sequali_amd import FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(18)
    U, n = 64, 16 * 300 + 3
    ids = [99999999, 100000000, 123456789012, 123456789012345678, 7]
    tl = _runs(rng, n, [1, 16, 33, 400], ids)
    buf, metas = _batch(rng, tl, U, qual_hi=60)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, gp = QCMetrics(), PerTileQuality()
    r = _route_of(lambda: with_env(RIDE, lambda: (FusedPass(gq, None, gp).add_record_array(arr), gq.flush(), gp.flush())))
    assert r.split("+")[0] == "k_span<2,QCPT,uniform,both>" and "k_pt_fold" in r, r
    raw = np.frombuffer(buf, np.uint8)
    want_err = {t: np.zeros(U) for t in ids}
    want_cnt = {t: 0 for t in ids}
    for i, t in enumerate(tl):
        q0 = int(metas["record_start"][i]) + int(metas["qualities_offset"][i])
        want_err[t] += 10.0 ** (-(raw[q0:q0 + U].astype(np.float64) - 33.0) / 10.0)
        want_cnt[t] += 1
    got = gp.get_tile_counts()
    assert [t for t, _, _ in got] == sorted(t for t in ids if want_cnt[t])
    assert gp.number_of_reads == n and gp.skipped_reason is None
    for t, e, c in got:
        np.testing.assert_allclose(np.array(e), want_err[t], rtol=1e-6, atol=0, err_msg=f"tile {t}")
        assert int(c[U - 1]) == want_cnt[t] and int(c[0]) == want_cnt[t]    # reverse-cumulated: every read has U bases


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


# ---- the pass over PAIRS of arrays (sq_paired_add_batches / PairedPass): read 2's pass leaves the ends of its reads
# behind, read 1's pass scans against them (calculate_insert_size, _qcmodule.c:5667-5707), both carry PerTileQuality ----

def _pair_batches(rng, n, L1, L2, tiles):
    """pairs cut from fragments of every length around the read lengths (read-through into the adapters, overlaps, none),
    with the cases the scan has to get right: one and two mismatches under a needle, N and lower case (the comparison is
    case-insensitive for the prefilter and raw for the verdict, :5695-5704), half a needle where it does not belong"""
    from tests.test_gpu_vs_oracle import _revcomp
    ad1, ad2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"

    def rand(k):
        return rng.choice(LETTERS, size=k).tobytes().decode()
    n1, s1, q1, n2, s2, q2 = [], [], [], [], [], []
    for i in range(n):
        flen = 16 + (i * 7) % (L1 + L2) if i % 5 else int(rng.integers(16, L1 + L2 + 40))
        frag = rand(flen)
        r1 = (frag + ad1 + "G" * 300)[:L1]
        r2 = (_revcomp(frag) + ad2 + "G" * 300)[:L2]
        kind = i % 11
        if kind == 1:
            at = min(max(flen - 16 + 5, 0), L1 - 1)
            r1 = r1[:at] + ("A" if r1[at] != "A" else "C") + r1[at + 1:]
        elif kind == 2:
            for at in (min(max(flen - 16 + 2, 0), L1 - 1), min(max(flen - 16 + 9, 0), L1 - 1)):
                r1 = r1[:at] + ("A" if r1[at] != "A" else "C") + r1[at + 1:]
        elif kind == 3:
            r2 = r2[:3] + "N" + r2[4:]
        elif kind == 4:
            at = int(rng.integers(0, L1))
            r1 = r1[:at] + "N" + r1[at + 1:]
        elif kind == 5:
            r1 = r1.lower()
        elif kind == 6 and L1 >= 40:
            r1 = r1[:20] + _revcomp(r2[:16])[:8] + r1[28:]
        elif kind == 7:
            r2 = r2[:L2 - 5] + r2[L2 - 5:].lower()
        t = tiles[i]
        n1.append(f"M0:7:FCX:{1 + i % 4}:{t}:{1000 + i}:{L1} 1:N:0:ACGT")
        n2.append(f"M0:7:FCX:{1 + i % 4}:{t}:{1000 + i}:{L2} 2:N:0:ACGT")
        s1.append(r1[:L1]); q1.append((rng.integers(0, 60, size=L1) + 33).astype(np.uint8).tobytes().decode())
        s2.append(r2[:L2]); q2.append((rng.integers(0, 60, size=L2) + 33).astype(np.uint8).tobytes().decode())
    return oracle.make_batch(n1, s1, q1), oracle.make_batch(n2, s2, q2)


def _compare_insert_sizes(got, ref):
    np.testing.assert_array_equal(u64(got.insert_sizes()), ref.insert_sizes())
    assert got.adapters_read1() == ref.adapters_read1()
    assert got.adapters_read2() == ref.adapters_read2()
    assert got.number_of_adapters_read1 == ref.number_of_adapters_read1
    assert got.number_of_adapters_read2 == ref.number_of_adapters_read2
    assert got.total_reads == ref.total_reads


@pytest.mark.parametrize("L1,L2,n", [(150, 150, 16 * 300), (150, 150, 16 * 300 + 9), (151, 75, 16 * 280 + 3), (64, 200, 16 * 260),
                                     (16, 16, 4096), (250, 250, 16 * 257 + 1), (40, 31, 16 * 300), (100, 16, 16 * 256 + 15)])
def test_paired_pass_equals_the_five_calls(L1, L2, n):
    """two batches of pairs through PairedPass against the oracle's five modules: QCMetrics and PerTileQuality of both
    mates, InsertSizeMetrics (histogram, both adapter tables in slot order, the counters), with a small max_adapters so
    that the first-come cap is crossed; the route is asserted (two passes and the remainders' kernel, no k_isz_span)"""
    from sequali_amd import FastqRecordArrayView, InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(L1 * 1000 + L2 + n)
    ref = (oracle.QCMetrics(), oracle.PerTileQuality(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.InsertSizeMetrics(50))
    got = (QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics(50))
    pp = PairedPass(*got)
    nw1, nw2 = (L1 + 31) // 32, (L2 + 31) // 32
    for part in range(2):
        tiles = _runs(rng, n, [3, 16, 40, 700], [1101, 1102, 2203])
        (b1, m1), (b2, m2) = _pair_batches(rng, n, L1, L2, tiles)
        ref[0].add(b1, m1); ref[1].add(b1, m1); ref[2].add(b2, m2); ref[3].add(b2, m2); ref[4].add_pair(b1, m1, b2, m2)
        a1 = FastqRecordArrayView._from_buffer(b1, m1.copy())
        a2 = FastqRecordArrayView._from_buffer(b2, m2.copy())
        r = _route_of(lambda: with_env(RIDE, lambda: (pp.add_record_array_pair(a1, a2), got[0].flush(), got[2].flush(), got[4].insert_sizes())))
        want = [f"k_span<{nw2},QCPT_ends,uniform,both>", "k_pt_fold", f"k_span<{nw1},QCPT_scan,uniform,both>", "k_pt_fold", "k_isz_adapters<hist>"]
        parts = [p for p in r.split("+") if not p.startswith("k_pass")]     # (the records behind the last full span: QCMetrics by k_pass)
        assert parts[:5] == want and "k_isz_span" not in r, r
        compare_qc(ref[0], got[0], m1, a1)
        compare_qc(ref[2], got[2], m2, a2)
        _compare_pertile(got[1], ref[1])
        _compare_pertile(got[3], ref[3])
        _compare_insert_sizes(got[4], ref[4])


def test_paired_pass_falls_back_to_the_five_calls():
    """what the paired kernels do not take goes through the modules' own passes with the same results: reads of many
    lengths, a mate shorter than the 16 bases of a needle, SQ_PT_FUSED=0, modules left out"""
    from sequali_amd import FastqRecordArrayView, InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(11)
    n = 16 * 280 + 2
    tiles = _runs(rng, n, [50, 900], [1101, 1102])
    (b1, m1), (b2, m2) = _pair_batches(rng, n, 120, 90, tiles)
    m1r = m1.copy()
    m1r["sequence_length"][::3] -= 7            # ragged read 1 (the qualities start where they did: only the length moves)
    (c1, k1), (c2, k2) = _pair_batches(rng, n, 150, 12, tiles)
    cases = [(b1, m1r, b2, m2, RIDE), (c1, k1, c2, k2, RIDE), (b1, m1, b2, m2, {"SQ_PT_FUSED": "0"})]
    for x1, y1, x2, y2, env in cases:
        ref = (oracle.QCMetrics(), oracle.PerTileQuality(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.InsertSizeMetrics())
        ref[0].add(x1, y1); ref[1].add(x1, y1); ref[2].add(x2, y2); ref[3].add(x2, y2); ref[4].add_pair(x1, y1, x2, y2)
        got = (QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics())
        a1 = FastqRecordArrayView._from_buffer(x1, y1.copy())
        a2 = FastqRecordArrayView._from_buffer(x2, y2.copy())
        r = _route_of(lambda: with_env(env, lambda: (PairedPass(*got).add_record_array_pair(a1, a2), got[0].flush(), got[2].flush(), got[4].insert_sizes())))
        assert "QCPT_scan" not in r, r
        compare_qc(ref[0], got[0], y1, a1)
        compare_qc(ref[2], got[2], y2, a2)
        _compare_pertile(got[1], ref[1])
        _compare_pertile(got[3], ref[3])
        _compare_insert_sizes(got[4], ref[4])
    # modules left out: QCMetrics of read 1 and InsertSizeMetrics alone
    ref_q, ref_z = oracle.QCMetrics(), oracle.InsertSizeMetrics()
    ref_q.add(b1, m1); ref_z.add_pair(b1, m1, b2, m2)
    q, z = QCMetrics(), InsertSizeMetrics()
    a1 = FastqRecordArrayView._from_buffer(b1, m1.copy())
    a2 = FastqRecordArrayView._from_buffer(b2, m2.copy())
    with_env(RIDE, lambda: (PairedPass(q, None, None, None, z).add_record_array_pair(a1, a2), q.flush()))
    compare_qc(ref_q, q, m1, a1)
    _compare_insert_sizes(z, ref_z)


def test_paired_pass_on_device_batches_by_tile():
    """the bench's config 3 at a small size: device-generated pairs in the order a sequencer writes"""
    from sequali_amd import InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics, synth
    n, first = 500_000, 3 * 65536 - 1000
    d1 = synth.device_array(synth.ILLUMINA_BY_TILE, first, n)
    d2 = synth.device_array(synth.ILLUMINA_R2_BY_TILE, first, n)
    b1, m1 = d1._batch.download()
    b2, m2 = d2._batch.download()
    ref = (oracle.QCMetrics(), oracle.PerTileQuality(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.InsertSizeMetrics())
    ref[0].add(b1, m1); ref[1].add(b1, m1); ref[2].add(b2, m2); ref[3].add(b2, m2); ref[4].add_pair(b1, m1, b2, m2)
    got = (QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics())
    r = _route_of(lambda: with_env(RIDE, lambda: (PairedPass(*got).add_record_array_pair(d1, d2), got[0].flush(), got[2].flush(), got[4].insert_sizes())))
    assert r == "k_span<5,QCPT_ends,uniform,both>+k_pt_fold+k_span<5,QCPT_scan,uniform,both>+k_pt_fold+k_isz_adapters<hist>", r
    compare_qc(ref[0], got[0], m1, d1)
    compare_qc(ref[2], got[2], m2, d2)
    _compare_pertile(got[1], ref[1])
    _compare_pertile(got[3], ref[3])
    _compare_insert_sizes(got[4], ref[4])


def test_paired_pass_through_the_parsers_at_the_default_buffer_size():
    """the reference's call pattern (__main__.py:279-306): two FastqParsers at 128 KiB, read 2 by read(len(array 1)), one
    PairedPass call per pair of arrays (a few hundred reads each).  The arrays are staged by the parsers' blocks and counted
    a block at a time; the results are those of the oracle on the whole files"""
    import io
    from sequali_amd import FastqParser, InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics, synth
    from tests.helpers import split_fastq
    n = 120_000
    t1, t2 = synth.host_records(synth.ILLUMINA_BY_TILE, 65536 - 40_000, n)[0], synth.host_records(synth.ILLUMINA_R2_BY_TILE, 65536 - 40_000, n)[0]
    (b1, m1), (b2, m2) = split_fastq(t1), split_fastq(t2)
    ref = (oracle.QCMetrics(), oracle.PerTileQuality(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.InsertSizeMetrics())
    ref[0].add(b1, m1); ref[1].add(b1, m1); ref[2].add(b2, m2); ref[3].add(b2, m2); ref[4].add_pair(b1, m1, b2, m2)
    for env in (RIDE, {}):
        got = (QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics())

        def run():
            pp = PairedPass(*got)
            r1, r2 = FastqParser(io.BytesIO(t1)), FastqParser(io.BytesIO(t2))
            arrays = 0
            for a1 in r1:
                a2 = r2.read(len(a1))
                assert a1.is_mate(a2)
                pp.add_record_array_pair(a1, a2)
                arrays += 1
            assert arrays > 100
            got[0].flush(); got[2].flush(); got[4].insert_sizes()
        with_env(env, run)
        for g, r in ((got[0], ref[0]), (got[2], ref[2])):
            assert g.number_of_reads == r.number_of_reads == n
            np.testing.assert_array_equal(u64(g.base_count_table()), r.base_count_table())
            np.testing.assert_array_equal(u64(g.phred_count_table()), r.phred_count_table())
            np.testing.assert_array_equal(u64(g.gc_content()), r.gc_content())
            np.testing.assert_array_equal(u64(g.phred_scores()), r.phred_scores())
        _compare_pertile(got[1], ref[1])
        _compare_pertile(got[3], ref[3])
        _compare_insert_sizes(got[4], ref[4])


@pytest.mark.parametrize("bad_in", [1, 2])
def test_an_invalid_phred_character_through_the_fused_passes(bad_in):
    """A byte that is no phred character (:2073-2075, :2102-2105) in read 1 / in read 2 of a batch of pairs, through
    PairedPass (k_span<QCPT_ends> / <QCPT_scan>) and through FusedPass(QCMetrics, None, PerTileQuality): the flush
    raises the reference's ValueError and the QCMetrics object that met the byte holds what the reference's had counted
    when it raised -- the reads in front of it and the part of the bad read in front of the byte's group of four."""
    from sequali_amd import FastqRecordArrayView, FusedPass, InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(90 + bad_in)
    U, n, bad = 100, 16 * 300 + 7, 16 * 111 + 5
    tiles = _runs(rng, n, [40, 300, 1000], [1101, 1102, 2205])
    b1, m1 = _batch(rng, tiles, U)
    b2, m2 = _batch(rng, tiles, U, name_of=lambda i, t: f"M0:7:FCX:{1 + i % 4}:{t}:{1000 + i}:{U} 2:N:0:ACGT")
    bufs = {1: bytearray(b1), 2: bytearray(b2)}
    metas = {1: m1, 2: m2}
    mb = metas[bad_in][bad]
    bufs[bad_in][int(mb["record_start"]) + int(mb["qualities_offset"]) + 57] = 0x7F
    b1, b2 = bytes(bufs[1]), bytes(bufs[2])
    ref, ref_metas = oracle.QCMetrics(), metas[bad_in].copy()
    with pytest.raises(ValueError):
        ref.add(b1 if bad_in == 1 else b2, ref_metas)

    def check(q):
        assert q.number_of_reads == ref.number_of_reads and q.max_length == ref.max_length
        for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                     "end_anchored_phred_count_table", "gc_content", "phred_scores"):
            np.testing.assert_array_equal(u64(getattr(q, name)()), getattr(ref, name)(), err_msg=name)

    # the five calls of the driver loop as one
    q1, p1, q2, p2, z = QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics()
    a1, a2 = FastqRecordArrayView._from_buffer(b1, m1.copy()), FastqRecordArrayView._from_buffer(b2, m2.copy())
    bad_q = q1 if bad_in == 1 else q2

    def paired():      # (arrays of this size are staged: the pass runs when somebody asks for the state)
        PairedPass(q1, p1, q2, p2, z).add_record_array_pair(a1, a2)
        with pytest.raises(ValueError, match="Not a valid phred character"):
            bad_q.flush()
    r = _route_of(paired)
    assert "k_span<4,QCPT" in r, r
    check(bad_q)
    # QCMetrics + PerTileQuality of the bad mate alone
    q, p = QCMetrics(), PerTileQuality()
    arr = FastqRecordArrayView._from_buffer(b1 if bad_in == 1 else b2, metas[bad_in].copy())
    def fused():
        FusedPass(q, None, p).add_record_array(arr)
        with pytest.raises(ValueError, match="Not a valid phred character"):
            q.flush()
    r = _route_of(fused)
    assert r.split("+")[0] == "k_span<4,QCPT,uniform,both>", r
    check(q)
