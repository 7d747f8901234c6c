"""Edges of k_span / k_ptspan that the sweeps of test_gpu_vs_oracle.py only meet by chance:
adapters of the longest length the quarter-per-lane automaton takes, ending on the first and on
the last base of every lane's quarter; three and more hits inside one quarter (the second,
base-by-base walk); PerTileQuality tables that leave k_ptspan fewer waves or do not fit LDS at
all.  Reference semantics: _qcmodule.c:2549-2591 (adapters), :2657-2668 (first hit per adapter and
read), :3046 (tile table growth).  Needs a GPU."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import with_env
from tests.test_gpu_vs_oracle import compare_qc, u64

pytestmark = pytest.mark.gpu

LETTERS = np.frombuffer(b"ACGT", np.uint8)
A13, A12 = "ACGGTCATTGCAC", "TGACCGTTAGCA"


def _compare_adapters(ga, ra):
    total = 0
    for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
        np.testing.assert_array_equal(u64(f), fr)
        np.testing.assert_array_equal(u64(r), rr)
        total += int(fr.sum())
    return total


def _quarter(U):
    """positions per lane quarter of k_span for reads of U bases: 4 (2 NW + 1)"""
    return 4 * (2 * ((U + 31) // 32) + 1)


@pytest.mark.parametrize("U", [150, 97, 160, 64, 200, 224])
@pytest.mark.parametrize("route", ["uniform", "sorted", "unsplit"])
def test_adapter_ending_on_every_quarter_seam(U, route):
    """a 13-character adapter (the most SPAN_W4 = 3 dwords of restart in front of a quarter cover) and
    a 12-character one, ending on the first base, the last base and next to both seams of every
    lane quarter; the same read with two plants; uniform batch, length-sorted route (a few reads of
    other lengths make the batch ragged) and the kernel with one wave for both streams"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(31 * U)
    qs = _quarter(U)
    ends = sorted({e for c in range(5) for e in (qs * c - 2, qs * c - 1, qs * c, qs * c + 1, qs * c + qs - 1) if 0 <= e < U} | {U - 1, 12, 11})
    names, seqs, quals = [], [], []

    def add(s):
        names.append(f"r{len(names)}")
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=len(s)) + 33).astype(np.uint8).tobytes().decode())

    for rep in range(3):
        for ad in (A13, A12):
            for e in ends:
                if e < len(ad) - 1:
                    continue
                s = rng.choice(LETTERS, size=U).tobytes().decode()
                at = e - len(ad) + 1
                s = s[:at] + ad + s[at + len(ad):]
                if rep == 2:   # a second plant of the other adapter somewhere else
                    other = A12 if ad is A13 else A13
                    at2 = int(rng.integers(0, U - len(other) + 1))
                    if at2 + len(other) <= at or at2 >= at + len(ad):
                        s = s[:at2] + other + s[at2 + len(other):]
                add(s)
    while len(names) % 16 or len(names) < 256:
        add(rng.choice(LETTERS, size=U).tobytes().decode())
    if route == "sorted":
        for L in (U - 1, U - 7, max(13, U // 2), 13, 1):
            for _ in range(5):
                s = rng.choice(LETTERS, size=L).tobytes().decode()
                if L >= 13:
                    s = s[:L - 13] + A13
                add(s)
    buf, metas = oracle.make_batch(names, seqs, quals)
    probes = [A13, A12]
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, ga = QCMetrics(), AdapterCounter(probes)
    env = {"uniform": {}, "sorted": {"SQ_SPAN_SORTED": "1"}, "unsplit": {"SQ_SPAN_SPLIT": "0"}}[route]
    nw = (U + 31) // 32
    if nw <= 2:
        env = dict(env, SQ_SPAN_SHORT="1")   # (batches of one read length of up to 64 bases take k_wide by default since round 5: the k_span builds are what is tested here)
    from tests.test_gpu_vs_oracle import _route_of
    r = _route_of(lambda: with_env(env, lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush())))
    if route == "uniform":
        assert r.split("+")[0] == f"k_span<{nw},AD,uniform,split>", r
    elif route == "sorted":
        assert any(p.startswith(f"k_span<{nw},AD,sorted,") for p in r.split("+")), r
    elif nw <= 5:
        assert r.split("+")[0] == f"k_span<{nw},AD,uniform,both>", r
    compare_qc(rq, gq, metas, arr)
    assert _compare_adapters(ga, ra) >= 3 * len(ends)


@pytest.mark.parametrize("U,rate", [(150, 1.0), (150, 0.05), (96, 0.3)])
def test_three_and_more_hits_inside_one_quarter(U, rate):
    """a lane keeps its first two matches of a span in registers; a third one sends the wave through
    the base-by-base walk of its quarters.  Reads with three to five short adapters (distinct ones
    and repeats: only the first hit of an adapter counts, :2657-2668) inside ONE quarter, at a
    controlled share of the reads"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(U + int(100 * rate))
    probes = ["ACGTT", "GGCA", "TTAGC", "CATG", "GATTACA"]
    qs = _quarter(U)
    names, seqs, quals = [], [], []
    n = 16 * 40
    for i in range(n):
        s = list(rng.choice(LETTERS, size=U).tobytes().decode())
        if rng.random() < rate:
            c = int(rng.integers(0, (U + qs - 1) // qs))
            lo, hi = qs * c, min(qs * (c + 1), U)
            at = lo
            for _ in range(int(rng.integers(3, 6))):
                w = probes[int(rng.integers(0, len(probes)))]
                if at + len(w) > hi:
                    break
                s[at:at + len(w)] = w
                at += len(w) + int(rng.integers(0, 3))
        names.append(f"h{i}")
        seqs.append("".join(s))
        quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    for env in ({}, {"SQ_SPAN_SPLIT": "0"}):
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(probes)
        with_env(env, lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush()))
        compare_qc(rq, gq, metas, arr)
        _compare_adapters(ga, ra)


def _tiled_batch(rng, n, U, tiles):
    names, seqs, quals = [], [], []
    for i in range(n):
        t = tiles[int(rng.integers(0, len(tiles)))]
        names.append(f"M:1:F:{i % 4}:{t}:{i}:{U} 1:N:0:X")
        seqs.append(rng.choice(LETTERS, size=U).tobytes().decode())
        quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
    return oracle.make_batch(names, seqs, quals)


@pytest.mark.parametrize("U,ntiles", [(150, 90), (150, 100), (150, 110), (150, 130), (150, 700), (250, 96), (250, 40)])
def test_pertile_tables_that_squeeze_k_ptspan(U, ntiles):
    """k_ptspan keeps [tiles][U] doubles in LDS: 96 tiles of 150 positions leave 7 waves, more
    tiles fewer (sq_ptspan_launch's waves-- loop), and from some count on the table does not fit
    and k_ptq (behind a sort by tile) takes the batch; a NovaSeq lane has several hundred tiles.
    Counts exact, sums within 1e-6 (:3189-3220)"""
    from sequali_amd import FastqRecordArrayView, PerTileQuality
    rng = np.random.default_rng(U * 1000 + ntiles)
    tiles = [1101 + 7 * k for k in range(ntiles)]
    n = max(16 * 64, 4 * ntiles)
    buf, metas = _tiled_batch(rng, n, U, tiles)
    ref = oracle.PerTileQuality()
    ref.add(buf, metas)
    got = PerTileQuality()
    got.add_record_array(FastqRecordArrayView._from_buffer(buf, metas.copy()))
    rt, gt = ref.get_tile_counts(), got.get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (t, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6, err_msg=f"tile {t}")
        np.testing.assert_array_equal(u64(c), cr, err_msg=f"tile {t}")


@pytest.mark.parametrize("raw", [0x80, 0xC8, 0xFF, 0x7F])
@pytest.mark.parametrize("ragged", [False, True])
def test_quality_bytes_from_128_on_are_no_phred_characters(raw, ragged):
    """BAM stores qualities as bytes and the decoder adds 33: raw 95 .. 222 reach the pass as bytes
    >= 128.  The reference raises 'Not a valid phred character' (:2073-2075, :2102-2105); every
    record in front of the bad one stays counted, the object goes on counting.  Through k_span
    (uniform) and the length-sorted route, whose filler rows use byte 0x80 themselves"""
    from sequali_amd import FastqRecordArrayView, QCMetrics
    rng = np.random.default_rng(raw)
    U, n, bad = 100, 16 * 20, 16 * 11 + 5
    names = [f"r{i}" for i in range(n)]
    lens = [U - (i % 3 if ragged else 0) for i in range(n)]
    seqs = [rng.choice(LETTERS, size=L).tobytes() for L in lens]
    quals = [(rng.integers(0, 94, size=L) + 33).astype(np.uint8) for L in lens]
    quals[bad][lens[bad] // 2] = raw
    # the batch by hand: make_batch takes str
    parts, metas, pos = [], np.zeros(n, dtype=oracle.META_DTYPE), 0
    for i in range(n):
        rec = b"@" + names[i].encode() + b"\n" + seqs[i] + b"\n+\n" + quals[i].tobytes() + b"\n"
        nl = len(names[i])
        metas[i] = (pos + 1, nl, nl + 1, lens[i], nl + 1 + lens[i] + 3, nl + 1 + 2 * lens[i] + 3, 0, 0.0)
        parts.append(rec)
        pos += len(rec)
    buf = b"".join(parts)
    ref, ref_metas = oracle.QCMetrics(), metas.copy()
    with pytest.raises(ValueError):   # the oracle leaves the reference's state behind its exception
        ref.add(buf, ref_metas)
    env = {"SQ_SPAN_SORTED": "1"} if ragged else {}
    m = QCMetrics()
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())

    def run():
        m.add_record_array(arr)
        with pytest.raises(ValueError, match="Not a valid phred character"):
            m.flush()
    with_env(env, run)
    assert m.number_of_reads == ref.number_of_reads and m.max_length == ref.max_length
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        np.testing.assert_array_equal(u64(getattr(m, name)()), getattr(ref, name)(), err_msg=name)
    # and the object goes on counting
    more = FastqRecordArrayView._from_buffer(buf, metas[:bad].copy())
    ref.add(buf, metas[:bad].copy())
    with_env(env, lambda: (m.add_record_array(more), m.flush()))
    np.testing.assert_array_equal(u64(m.phred_count_table()), ref.phred_count_table())
    assert m.number_of_reads == ref.number_of_reads


@pytest.mark.timeout(700)
def test_fuzz_thirty_iterations():
    """scripts/fuzz.py (random batches, module parameters and batch splits through every module against the oracle) as
    part of the suite: 30 iterations of a fixed seed.  Ten minutes at most: in round 4 the run was cut off at 300 s by
    the suite's default timeout with nothing to show for it; the script prints a line per iteration, and what it had
    printed when the time ran out names the iteration to look at"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, "-u", os.path.join(root, "scripts", "fuzz.py"), "30", "4"], capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        raise AssertionError("scripts/fuzz.py 30 4 did not finish in 600 s; its output so far:\n" + out[-3000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "failures: 0" in r.stdout


def test_few_very_long_reads_with_more_than_64_adapters():
    """k_span<LONG> with a second adapter group: the reads-per-segment table of the long route has a scratch slot of its
    own (it used to share the one P.order lives in, which the passes of further adapter groups still walk) -- few
    reads, one of them long enough for more segments than there are reads"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(99)
    n = 4200
    lens = rng.integers(300, 1500, size=n)
    lens[17] = 1_300_000
    lens[18] = 70_000
    # the first group of 64 must fit k_span's automaton (336 states of the two-character one; all 64 three-letter words
    # alone make 409, counted on the host with sq_adapter_automaton_tables): twelve letters, the first nine shared: 244
    adapters = ["GATTACAGA" + a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT"]
    while len(adapters) < 70:
        w = rng.choice(LETTERS, size=int(rng.integers(8, 13))).tobytes().decode()
        if w not in adapters:
            adapters.append(w)
    names, seqs, quals = [], [], []
    for i, L in enumerate(lens):
        L = int(L)
        s = rng.choice(LETTERS, size=L).tobytes().decode()
        for _ in range(2):
            w = adapters[int(rng.integers(0, len(adapters)))]
            at = int(rng.integers(0, L - len(w) + 1))
            s = s[:at] + w + s[at + len(w):]
        names.append(f"n{i}")
        seqs.append(s)
        quals.append((rng.integers(0, 60, size=L) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(adapters)
    rq.add(buf, metas)
    ra.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, ga = QCMetrics(), AdapterCounter(adapters)
    from tests.test_gpu_vs_oracle import _route_of
    r = _route_of(lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush()))
    assert "k_span<8,AD,long>" in r, r
    compare_qc(rq, gq, metas, arr)
    assert _compare_adapters(ga, ra) > n


A25, A20, A14 = "ACGGTCATTGCACTTAGGCATCGAT", "TGACCGTTAGCAGGATCCTA", "GTTACCAGTCAGGA"


@pytest.mark.parametrize("U", [150, 97, 200, 224])
@pytest.mark.parametrize("route", ["uniform", "sorted", "unsplit"])
def test_adapters_of_14_to_25_characters_on_every_quarter_seam(U, route):
    """adapters of 25, 20 and 14 characters (AdapterCounter takes up to 64, _qcmodule.c:2549-2591) through the builds of
    k_span whose automaton is restarted six dwords in front of a lane's quarter (csrc/sq_span_w6.hip, SQ_SPAN_W6=1):
    planted so that they end on the first base, the last base and next to both seams of every quarter, and start up to
    24 positions inside the quarter in front; a second plant elsewhere in a third of the reads.  Without the switch the
    same batch takes k_wide (the cross-check)."""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    from tests.test_gpu_vs_oracle import _route_of
    rng = np.random.default_rng(77 * U)
    qs = _quarter(U)
    ends = sorted({e for c in range(5) for e in (qs * c - 2, qs * c - 1, qs * c, qs * c + 1, qs * c + 11, qs * c + 23, qs * c + qs - 1) if 0 <= e < U} | {U - 1, 24, 13})
    probes = [A25, A20, A14]
    names, seqs, quals = [], [], []

    def add(s):
        names.append(f"r{len(names)}")
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=len(s)) + 33).astype(np.uint8).tobytes().decode())

    for rep in range(3):
        for ad in probes:
            for e in ends:
                if e < len(ad) - 1:
                    continue
                s = rng.choice(LETTERS, size=U).tobytes().decode()
                at = e - len(ad) + 1
                s = s[:at] + ad + s[at + len(ad):]
                if rep == 2:
                    other = probes[(probes.index(ad) + 1) % 3]
                    at2 = int(rng.integers(0, U - len(other) + 1))
                    if at2 + len(other) <= at or at2 >= at + len(ad):
                        s = s[:at2] + other + s[at2 + len(other):]
                add(s)
    while len(names) % 16 or len(names) < 256:
        add(rng.choice(LETTERS, size=U).tobytes().decode())
    if route == "sorted":
        for L in (U - 1, U - 7, max(70, U // 2), 70, 65):
            for _ in range(5):
                s = rng.choice(LETTERS, size=L).tobytes().decode()
                add(s[:L - 25] + A25)
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    nw = (U + 31) // 32
    for w6 in (True, False):
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(probes)
        env = {"uniform": {}, "sorted": {"SQ_SPAN_SORTED": "1"}, "unsplit": {"SQ_SPAN_SPLIT": "0"}}[route]
        env = dict(env, SQ_SPAN_W6="1" if w6 else "0")
        r = _route_of(lambda: with_env(env, lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush())))
        if w6 and route == "uniform":
            assert r.split("+")[0] == f"k_span<{nw},AD,uniform,split,w6>", r
        elif w6 and route == "sorted":
            assert any(p.startswith(f"k_span<{nw},AD,sorted,") and p.endswith(",w6>") for p in r.split("+")), r
        elif w6 and nw <= 5:
            assert r.split("+")[0] == f"k_span<{nw},AD,uniform,both,w6>", r
        elif not w6:
            assert "w6" not in r and "k_span<" not in r.split("+")[0], r     # k_wide / k_pass as before
        compare_qc(rq, gq, metas, arr)
        assert _compare_adapters(ga, ra) >= 3 * len(ends)


@pytest.mark.parametrize("U", [150, 33, 97, 200, 250])
@pytest.mark.parametrize("route", ["uniform", "sorted", "unsplit", "pertile"])
def test_bases_a_sequencer_does_not_write(U, route):
    """k_span classifies the bytes sequencers write (A C G T N, either case) with one v_perm per dword and takes
    cls6_of_dword (any byte, NUCLEOTIDE_TO_INDEX :1748-1763) only for a span that holds something else: reads of the
    usual letters, reads with ONE other byte (first base, last base, next to a lane seam; the neighbours of the letters
    in the low three bits -- Q ! 1 for A, S # for C, W ' for G, D $ for T, F & for N --, IUPAC codes, '.', '-', '*'),
    reads of arbitrary printable bytes, mixed so that spans of 16 reads hold none, one and many of them; the text
    behind a read's last base (newline, '+', qualities) must not count as unusual, nor as anything else"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(7 * U + len(route))
    usual = np.frombuffer(b"ACGTNacgtn", np.uint8)
    odd = np.frombuffer(b"Q!1S#W'D$F&RYKMBVHUX.-*@`~0", np.uint8)
    probes = [A12[:min(U, 12)], "GGNCA"[:min(U, 5)]]
    names, seqs, quals = [], [], []
    n = 16 * 36 + 5
    qs = _quarter(U)
    for i in range(n):
        L = U if route != "sorted" or i % 5 else max(1, U - 1 - i % 9)
        block = (i // 16) % 4          # spans: all usual / one odd byte in one read / one odd byte in some reads / anything
        s = rng.choice(usual, size=L, p=[.2, .2, .2, .2, .02, .04, .04, .04, .04, .02])
        if block == 1 and i % 16 == 5 or block == 2 and rng.random() < 0.4:
            at = [0, L - 1, min(L - 1, qs - 1), min(L - 1, qs), int(rng.integers(0, L))][int(rng.integers(0, 5))]
            s[at] = odd[int(rng.integers(0, len(odd)))]
        elif block == 3:
            s = rng.integers(33, 127, size=L).astype(np.uint8)
        s = s.tobytes().decode()
        if L >= 12 and i % 4 == 0:
            at = int(rng.integers(0, L - 11))
            s = s[:at] + A12 + s[at + 12:]
        names.append(f"M:1:F:1:{1101 + i // 200}:5:{i}" if route == "pertile" else "r" * (1 + i % 50))
        seqs.append(s)
        quals.append((rng.integers(0, 42, size=L) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra, rp = oracle.QCMetrics(), oracle.AdapterCounter(probes), oracle.PerTileQuality()
    rq.add(buf, metas)
    ra.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, ga, gp = QCMetrics(), AdapterCounter(probes), PerTileQuality()
    env = {"uniform": {}, "sorted": {"SQ_SPAN_SORTED": "1"}, "unsplit": {"SQ_SPAN_SPLIT": "0"}, "pertile": {}}[route]
    if (U + 31) // 32 <= 2:
        env = dict(env, SQ_SPAN_SHORT="1")
    from tests.test_gpu_vs_oracle import _route_of
    if route == "pertile":
        rp.add(buf, metas)
        r = _route_of(lambda: with_env(env, lambda: (FusedPass(gq, None, gp).add_record_array(arr), gq.flush())))
        assert "QCPT" in r, r
    else:
        r = _route_of(lambda: with_env(env, lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush())))
        if route != "unsplit" or (U + 31) // 32 <= 5:   # (one wave for both streams with the automaton: up to 5 windows)
            assert r.startswith("k_span") and "k_span<" in r, r
    compare_qc(rq, gq, metas, arr)
    if route == "pertile":
        rt, gt = rp.get_tile_counts(), gp.get_tile_counts()
        assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
        for (t, e, cnt), (_, er, cr) in zip(gt, rt):
            np.testing.assert_allclose(np.array(e), er, rtol=1e-6, err_msg=f"tile {t}")
            np.testing.assert_array_equal(u64(cnt), cr, err_msg=f"tile {t}")
    else:
        _compare_adapters(ga, ra)


def test_long_reads_with_bases_a_sequencer_does_not_write():
    """the same through k_span<LONG> (segments of long reads): spans of full segments with the usual letters take the short
    way, a span with another byte or with a row that ends inside its segment cls6_of_dword"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(77)
    usual = np.frombuffer(b"ACGTNacgtn", np.uint8)
    odd = np.frombuffer(b"Q!1S#W'D$F&RYKMBVHUX.-*@`~0", np.uint8)
    probes = [A12, "GGNCA"]
    names, seqs, quals = [], [], []
    n = 4200   # (the long route sorts its reads by length: batches of 4096 reads and more)
    for i in range(n):
        L = int(rng.integers(300, 1500)) if i % 7 else int(rng.integers(257, 300)) if i % 14 else int(rng.integers(1500, 6000))
        s = rng.choice(usual, size=L, p=[.2, .2, .2, .2, .02, .04, .04, .04, .04, .02])
        if i % 3 == 0:
            for at in ([0, L - 1, 255, 256, 257] + [int(x) for x in rng.integers(0, L, size=3)])[:int(rng.integers(1, 9))]:
                s[min(at, L - 1)] = odd[int(rng.integers(0, len(odd)))]
        elif i % 11 == 5:
            s = rng.integers(33, 127, size=L).astype(np.uint8)
        s = s.tobytes().decode()
        if i % 4 == 0:
            at = int(rng.integers(0, L - 11))
            s = s[:at] + A12 + s[at + 12:]
        names.append(f"read{i}")
        seqs.append(s)
        quals.append((rng.integers(0, 42, size=L) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq, ga = QCMetrics(), AdapterCounter(probes)
    from tests.test_gpu_vs_oracle import _route_of
    r = _route_of(lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush()))
    assert "long>" in r, r
    compare_qc(rq, gq, metas, arr)
    _compare_adapters(ga, ra)
