"""HIP path vs the CPU oracle on seeded inputs the golden files do not cover:
ragged lengths, every byte class, several adapter group shapes, long reads,
device-generated batches.  Needs a GPU."""
import warnings

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def u64(a):
    return np.array(a, dtype=np.uint64)


def random_batch(rng, n, max_len, illumina_names=True, alphabet=b"ACGTNacgtnRYKM-", qmax=93,
                 splice=()):
    from sequali_amd import FastqRecordArrayView
    names, seqs, quals = [], [], []
    alpha = np.frombuffer(alphabet, np.uint8)
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        if rng.random() < 0.1:
            L = int(rng.integers(0, 8))
        w = np.ones(len(alpha))
        w[:4] = 20
        s = rng.choice(alpha, size=L, p=w / w.sum()).tobytes().decode()
        if splice and rng.random() < 0.5:   # plant one or two of the given words
            for _ in range(int(rng.integers(1, 3))):
                word = splice[int(rng.integers(0, len(splice)))]
                if rng.random() < 0.3:
                    word = word.lower()
                if len(word) <= L:
                    at = int(rng.integers(0, L - len(word) + 1))
                    s = s[:at] + word + s[at + len(word):]
        q = (rng.integers(0, qmax + 1, size=L) + 33).astype(np.uint8).tobytes().decode()
        tile = int(rng.choice([1101, 1102, 2205, 7, 99239]))
        names.append(f"M:1:F:{i % 4}:{tile}:{i}:{L} 1:N:0:X" if illumina_names else f"read{i} x")
        seqs.append(s)
        quals.append(q)
    buf, metas = oracle.make_batch(names, seqs, quals)
    return buf, metas, FastqRecordArrayView._from_buffer(buf, metas.copy())


def compare_qc(ref, got, metas_ref, arr):
    assert got.number_of_reads == ref.number_of_reads
    assert got.max_length == ref.max_length
    np.testing.assert_array_equal(u64(got.base_count_table()), ref.base_count_table())
    np.testing.assert_array_equal(u64(got.phred_count_table()), ref.phred_count_table())
    np.testing.assert_array_equal(u64(got.end_anchored_base_count_table()), ref.end_anchored_base_count_table())
    np.testing.assert_array_equal(u64(got.end_anchored_phred_count_table()), ref.end_anchored_phred_count_table())
    np.testing.assert_array_equal(u64(got.gc_content()), ref.gc_content())
    np.testing.assert_array_equal(u64(got.phred_scores()), ref.phred_scores())
    np.testing.assert_array_equal(arr.accumulated_error_rates().view(np.uint64),
                                  metas_ref["accumulated_error_rate"].view(np.uint64))


@pytest.mark.parametrize("seed,n,max_len,ea", [(1, 2000, 151, 100), (2, 3000, 40, 100), (3, 700, 700, 15),
                                                (4, 300, 3000, 300), (5, 64, 64, 0), (6, 65, 63, 1000)])
def test_qc_ragged(seed, n, max_len, ea):
    from sequali_amd import QCMetrics
    rng = np.random.default_rng(seed)
    buf, metas, arr = random_batch(rng, n, max_len)
    ref, got = oracle.QCMetrics(ea), QCMetrics(ea)
    ref.add(buf, metas)
    got.add_record_array(arr)
    compare_qc(ref, got, metas, arr)


def test_qc_two_batches_growing_length():
    from sequali_amd import QCMetrics
    rng = np.random.default_rng(11)
    ref, got = oracle.QCMetrics(), QCMetrics()
    for max_len in (50, 400, 120):
        buf, metas, arr = random_batch(rng, 500, max_len)
        ref.add(buf, metas)
        got.add_record_array(arr)
    np.testing.assert_array_equal(u64(got.base_count_table()), ref.base_count_table())
    np.testing.assert_array_equal(u64(got.phred_count_table()), ref.phred_count_table())
    np.testing.assert_array_equal(u64(got.phred_scores()), ref.phred_scores())
    assert got.max_length == ref.max_length


def test_qc_invalid_phred_raises():
    from sequali_amd import FastqRecordArrayView, QCMetrics
    buf, metas = oracle.make_batch(["a", "b"], ["ACGT", "ACGTA"], ["IIII", "II II"])
    m = QCMetrics()
    m.add_record_array(FastqRecordArrayView._from_buffer(buf, metas))
    with pytest.raises(ValueError, match="Not a valid phred character:  "):
        m.base_count_table()


@pytest.mark.parametrize("U,bad_read,bad_pos", [(150, 70, 149), (150, 3, 0), (100, 191, 64), (70, 130, 67)])
def test_invalid_phred_in_a_batch_of_one_length(U, bad_read, bad_pos):
    """the deferred ValueError also comes out of the kernels that take batches of one read
    length (k_ring for QCMetrics alone, k_wide with the adapters): the bad byte in the chains,
    behind them (the 1-4 trailing qualities) or in the first chunk"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(U + bad_read)
    n = 64 * 3 + 10
    names = [f"r{i}" for i in range(n)]
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=U).tobytes().decode() for _ in range(n)]
    quals = ["I" * U for _ in range(n)]
    quals[bad_read] = quals[bad_read][:bad_pos] + " " + quals[bad_read][bad_pos + 1:]
    buf, metas = oracle.make_batch(names, seqs, quals)
    for with_adapters in (False, True):
        m, a = QCMetrics(), AdapterCounter(["ACGTACGTAC"])
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        if with_adapters:
            FusedPass(m, a).add_record_array(arr)
        else:
            m.add_record_array(arr)
        with pytest.raises(ValueError, match="Not a valid phred character:  "):
            m.base_count_table()


ADAPTER_SETS = [
    ["AGATCGGAAGAG", "TGGAATTCTCGG", "GATCGTCGGACT", "CTGTCTCTTATA", "GGGGGGGGGGGG", "AAAAAAAAAAAA"],
    ["ACG", "CGT", "A", "NN", "GTAC", "TTTTTTTT"],
    ["A" * 64, "C" * 64, "G" * 64, "ACGT" * 16, "N" * 5],
    [("ACGT" * 16)[i:i + 7 + i % 5] for i in range(70)],   # more than 64 adapters: two automatons
]


@pytest.mark.parametrize("which", range(len(ADAPTER_SETS)))
def test_adapter_ragged(which):
    from sequali_amd import AdapterCounter
    rng = np.random.default_rng(20 + which)
    adapters = ADAPTER_SETS[which]
    buf, metas, arr = random_batch(rng, 3000, 200, alphabet=b"ACGTNacgt", splice=adapters)
    ref, got = oracle.AdapterCounter(adapters), AdapterCounter(adapters)
    ref.add(buf, metas)
    got.add_record_array(arr)
    assert got.max_length == ref.max_length
    assert got.number_of_sequences == ref.number_of_sequences
    for (_, f, r), (_, fr, rr) in zip(got.get_counts(), ref.get_counts()):
        np.testing.assert_array_equal(u64(f), fr)
        np.testing.assert_array_equal(u64(r), rr)
    assert sum(int(f.sum()) for _, f, _ in ref.get_counts()) > 0


def test_pertile_ragged():
    from sequali_amd import PerTileQuality
    rng = np.random.default_rng(31)
    ref, got = oracle.PerTileQuality(), PerTileQuality()
    for n in (2000, 6000, 5000):   # >= 4096 records: the tile-sorted processing order
        buf, metas, arr = random_batch(rng, n, 151)
        ref.add(buf, metas)
        got.add_record_array(arr)
    assert got.number_of_reads == ref.number_of_reads
    assert got.max_length == ref.max_length
    assert got.skipped_reason is None
    for (t, e, c), (tr, er, cr) in zip(got.get_tile_counts(), ref.get_tile_counts()):
        assert t == tr
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6)   # summation order differs
        np.testing.assert_array_equal(u64(c), cr)


def test_pertile_reads_ordered_by_tile_are_walked_as_stored():
    """a batch whose tiles come in long runs (what a sequencer writes) skips the tile sort:
    stored order, one-tile groups counted without asking per row; 100 k device-generated
    reads in runs of 65536 (two tiles, the change in the middle of a group) and the same batch
    forced through the sorted walk give the oracle's tables"""
    from sequali_amd import FusedPass, PerTileQuality, QCMetrics, synth
    n = 100_000
    dev = synth.device_array(synth.ILLUMINA_BY_TILE, 0, n)
    buf, metas = dev._batch.download()
    rp, rq = oracle.PerTileQuality(), oracle.QCMetrics()
    rp.add(buf, metas)
    rq.add(buf, metas)
    want = {t: (e, c) for t, e, c in rp.get_tile_counts()}
    assert len(want) == 2
    for env in ({},):
        gq, gp = QCMetrics(), PerTileQuality()
        _with_env(env, lambda: (FusedPass(gq, None, gp).add_record_array(dev), gq.flush()))
        got = {t: (e, c) for t, e, c in gp.get_tile_counts()}
        assert set(got) == set(want)
        for t in want:
            np.testing.assert_array_equal(u64(got[t][1]), want[t][1])
            np.testing.assert_allclose(np.array(got[t][0]), want[t][0], rtol=1e-6, atol=0)
        np.testing.assert_array_equal(u64(gq.phred_count_table()), rq.phred_count_table())
        np.testing.assert_array_equal(u64(gq.base_count_table()), rq.base_count_table())


@pytest.mark.parametrize("env", [{}, {"SQ_OVERREP_CHAIN": "1"}], ids=["k_overrep_par", "k_overrep"])
def test_overrep_vs_oracle_with_cap_crossing(env):
    """the cap is crossed inside a batch (the first 700 distinct fragments in sampled-read, staging-SLOT order stay):
    through k_overrep_par (a lane's loads in flight together, the staging slots re-enacted in registers) and through
    k_overrep (a lane's fragments one after the other, the staging table itself)"""
    from sequali_amd import OverrepresentedSequences
    rng = np.random.default_rng(41)
    kw = dict(max_unique_fragments=700, sample_every=3, fragment_length=21)
    ref, got = oracle.OverrepresentedSequences(**kw), OverrepresentedSequences(**kw)

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(4):
                buf, metas, arr = random_batch(rng, 1500, 160, alphabet=b"ACGTACGTACGTACGTN")
                ref.add(buf, metas)
                got.add_record_array(arr)
                got.flush()
    _with_env(env, run)
    assert got.collected_unique_fragments == ref.collected_unique_fragments == 700
    assert got.total_fragments == ref.total_fragments
    assert got.sampled_sequences == ref.sampled_sequences
    assert got.sequence_counts() == ref.sequence_counts()
    assert got.overrepresented_sequences(0.001) == ref.overrepresented_sequences(0.001)


@pytest.mark.parametrize("k,start,end,max_len,cap", [(21, 100, 100, 160, 5000), (5, 40, 40, 90, 300), (7, 20, 60, 400, 900),
                                                     (3, 8, 40, 60, 64), (31, 100, 100, 1200, 2000), (5, 38, 42, 80, 200)])
def test_overrep_fragment_geometries(k, start, end, max_len, cap):
    """reads that repeat fragments inside themselves (short k-mers: a fragment twice in a read is counted once,
    :3588-3608), more fragments from one end than from the other, up to 16 fragments a read, N and other letters, caps
    crossed in the middle of a batch"""
    from sequali_amd import OverrepresentedSequences
    rng = np.random.default_rng(1000 * k + start)
    kw = dict(max_unique_fragments=cap, sample_every=1, fragment_length=k, bases_from_start=start, bases_from_end=end)
    ref, got = oracle.OverrepresentedSequences(**kw), OverrepresentedSequences(**kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):
            buf, metas, arr = random_batch(rng, 2000, max_len, alphabet=b"ACGTACGTACGTACGTNacgtR")
            ref.add(buf, metas)
            got.add_record_array(arr)
    assert got.collected_unique_fragments == ref.collected_unique_fragments
    assert got.total_fragments == ref.total_fragments
    assert got.sequence_counts() == ref.sequence_counts()


def test_overrep_whole_read_fragments():
    """bases_from_start/end = -1: staging tables larger than the register path"""
    from sequali_amd import OverrepresentedSequences
    rng = np.random.default_rng(42)
    kw = dict(sample_every=1, fragment_length=5, bases_from_start=-1, bases_from_end=-1)
    ref, got = oracle.OverrepresentedSequences(**kw), OverrepresentedSequences(**kw)
    buf, metas, arr = random_batch(rng, 400, 400, alphabet=b"ACGT")
    ref.add(buf, metas)
    got.add_record_array(arr)
    assert got.sequence_counts() == ref.sequence_counts()
    assert got.total_fragments == ref.total_fragments


def test_dedup_vs_oracle_rebuilds():
    from sequali_amd import DedupEstimator
    rng = np.random.default_rng(51)
    kw = dict(max_stored_fingerprints=300, front_sequence_offset=8, back_sequence_offset=0)
    ref, got = oracle.DedupEstimator(**kw), DedupEstimator(**kw)
    for _ in range(5):
        buf, metas, arr = random_batch(rng, 2500, 80, alphabet=b"ACGT")
        ref.add(buf, metas)
        got.add_record_array(arr)
    assert got._modulo_bits == ref._modulo_bits >= 3
    assert got.tracked_sequences == ref.tracked_sequences
    np.testing.assert_array_equal(u64(got.duplication_counts()), ref.duplication_counts())


def test_dedup_found_and_new_hashes_mixed_device_and_host_pieces():
    """The estimator's table in HBM (csrc/sq_ends.hip, dedup_process): device-generated batches that overlap by
    half (found and new hashes mixed), a table small enough to be rebuilt several times -- through the parallel steps
    on the device, the same run forced through the host's sequential loop (SQ_DEDUP_SEQUENTIAL=1), and the oracle"""
    from sequali_amd import DedupEstimator, synth
    from sequali_amd._lib import lib
    kw = dict(max_stored_fingerprints=150_000, front_sequence_offset=0, back_sequence_offset=0)
    # 140 k new hashes fit, the next batch finds 40 k of them and brings 100 k new ones: the table fills inside the
    # batch; then larger batches
    spans = [(0, 140_000), (100_000, 140_000), (0, 300_000), (150_000, 300_000), (0, 300_000)]
    batches = [synth.device_array(synth.ILLUMINA, first, n) for first, n in spans]
    ref = oracle.DedupEstimator(**kw)
    for b in batches:
        buf, metas = b._batch.download()
        ref.add(buf, metas)
    for env in ({}, {"SQ_DEDUP_SEQUENTIAL": "1"}):
        got = DedupEstimator(**kw)
        _with_env(env, lambda: [got.add_record_array(b) for b in batches])
        assert got._modulo_bits == ref._modulo_bits >= 1
        assert got.tracked_sequences == ref.tracked_sequences
        np.testing.assert_array_equal(u64(got.duplication_counts()), ref.duplication_counts())
        dev, host = lib().sq_dedup_device_pieces(got._h), lib().sq_dedup_host_pieces(got._h)
        assert (dev == 0 and host > 0) if env else dev > 0


def test_dedup_pairs_with_short_reads_stale_bytes():
    """pairs shorter than the fingerprint reuse bytes of the previous fingerprint (:4512)"""
    from sequali_amd import DedupEstimator
    rng = np.random.default_rng(52)
    kw = dict(max_stored_fingerprints=200, front_sequence_offset=0, back_sequence_offset=0)
    ref, got = oracle.DedupEstimator(**kw), DedupEstimator(**kw)
    for _ in range(3):
        b1, m1, a1 = random_batch(rng, 800, 20, alphabet=b"ACGT")
        b2, m2, a2 = random_batch(rng, 800, 20, alphabet=b"ACGT")
        ref.add_pair(b1, m1, b2, m2)
        got.add_record_array_pair(a1, a2)
    assert got._modulo_bits == ref._modulo_bits
    np.testing.assert_array_equal(u64(got.duplication_counts()), ref.duplication_counts())


@pytest.mark.timeout(120)
def test_dedup_batches_of_nothing_but_short_pairs():
    """6000 pairs of at most 5 bases (what scripts/fuzz.py 30 4 draws in its second iteration): no pair ever
    rewrites the whole 16-byte store, so each fingerprint shows bytes of several pairs in front -- over batch
    borders too.  The store is carried from short pair to short pair; walking back from each of them to the
    last long pair, as the tail did until round 4, took minutes here."""
    import time
    from sequali_amd import DedupEstimator
    rng = np.random.default_rng(53)
    kw = dict(max_stored_fingerprints=100, front_sequence_offset=0, back_sequence_offset=0)
    ref, got = oracle.DedupEstimator(**kw), DedupEstimator(**kw)
    t0 = time.time()
    for n, max_len in ((6000, 5), (3000, 9), (1, 3), (2000, 5)):      # 9: a long pair now and then
        b1, m1, a1 = random_batch(rng, n, max_len, alphabet=b"ACGT")
        b2, m2, a2 = random_batch(rng, n, max_len, alphabet=b"ACGT")
        ref.add_pair(b1, m1, b2, m2)
        got.add_record_array_pair(a1, a2)
    assert time.time() - t0 < 60
    assert got._modulo_bits == ref._modulo_bits
    assert got.tracked_sequences == ref.tracked_sequences
    np.testing.assert_array_equal(u64(got.duplication_counts()), ref.duplication_counts())


def test_insert_size_vs_oracle():
    from sequali_amd import InsertSizeMetrics, synth
    ref, got = oracle.InsertSizeMetrics(50), InsertSizeMetrics(50)
    for first in (0, 5000):
        a1 = synth.host_array(synth.ILLUMINA, first, 5000)
        a2 = synth.host_array(synth.ILLUMINA_R2, first, 5000)
        ref.add_pair(a1.obj, a1._metas, a2.obj, a2._metas)
        got.add_record_array_pair(a1, a2)
    np.testing.assert_array_equal(u64(got.insert_sizes()), ref.insert_sizes())
    assert got.adapters_read1() == ref.adapters_read1()
    assert got.adapters_read2() == ref.adapters_read2()
    assert got.number_of_adapters_read1 == ref.number_of_adapters_read1 > 0
    assert int(ref.insert_sizes()[1:].sum()) > 1000


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgtN", "TGCAtgcaN"))


@pytest.mark.parametrize("L1,L2,n", [(150, 150, 16 * 40 + 7), (100, 75, 16 * 25), (16, 16, 16 * 9 + 15),
                                     (33, 250, 16 * 12 + 1), (256, 40, 16 * 10 + 3), (40, 151, 16 * 8)])
def test_insert_size_pairs_of_one_read_length(L1, L2, n):
    """pairs whose reads have one length each go through k_isz_span (read 1 streamed through LDS,
    four lanes per pair, a candidate bit per window) + k_isz_adapters, the pairs behind the last
    full span of 16 through k_insert_size: inserts of every size from 16 up to longer than both
    reads (the match in every quarter of read 1 and at its very end, found by the head or only by
    the tail needle), one mismatch in the overlap (still a match), two (no match), an N in the
    needle of read 2 and in the window of read 1, lower case bases, a decoy that matches one needle half
    only; histogram, adapter remainders and their order against the oracle, and against
    k_insert_size alone (SQ_SPAN=0)"""
    from sequali_amd import FastqRecordArrayView, InsertSizeMetrics
    rng = np.random.default_rng(L1 * 1000 + L2)
    ad1, ad2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"

    def rand(k):
        return rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k).tobytes().decode()
    n1, s1, q1, n2, s2, q2 = [], [], [], [], [], []
    for i in range(n):
        flen = 16 + (i * 7) % (L1 + L2) if i % 5 else int(rng.integers(16, L1 + L2 + 40))
        frag = rand(flen)
        r1 = (frag + ad1 + "G" * 300)[:L1]
        r2 = (_revcomp(frag) + ad2 + "G" * 300)[:L2]
        kind = i % 11
        if kind == 1 and flen >= 16:      # one mismatch inside what the head needle covers in read 1
            at = min(flen - 16 + 5, L1 - 1)
            r1 = r1[:at] + ("A" if r1[at] != "A" else "C") + r1[at + 1:]
        elif kind == 2 and flen >= 16:    # two of them
            for at in (min(flen - 16 + 2, L1 - 1), min(flen - 16 + 9, L1 - 1)):
                r1 = r1[:at] + ("A" if r1[at] != "A" else "C") + r1[at + 1:]
        elif kind == 3:
            r2 = r2[:3] + "N" + r2[4:]
        elif kind == 4:
            at = int(rng.integers(0, L1))
            r1 = r1[:at] + "N" + r1[at + 1:]
        elif kind == 5:
            r1, r2 = r1.lower(), r2
        elif kind == 6 and L1 >= 40:      # half of the head needle somewhere it does not belong
            half = _revcomp(r2[:16])[:8]
            r1 = r1[:20] + half + r1[28:]
        n1.append(f"p{i}/1"); s1.append(r1[:L1]); q1.append("I" * L1)
        n2.append(f"p{i}/2"); s2.append(r2[:L2]); q2.append("I" * L2)
    b1, m1 = oracle.make_batch(n1, s1, q1)
    b2, m2 = oracle.make_batch(n2, s2, q2)
    ref = oracle.InsertSizeMetrics(50)
    ref.add_pair(b1, m1, b2, m2)
    ref.add_pair(b1, m1, b2, m2)
    assert int(ref.insert_sizes()[1:].sum()) > 0
    for env in ({}, {"SQ_SPAN": "0"}):
        got = InsertSizeMetrics(50)

        def run():
            for _ in range(2):
                got.add_record_array_pair(FastqRecordArrayView._from_buffer(b1, m1.copy()),
                                          FastqRecordArrayView._from_buffer(b2, m2.copy()))
            got.insert_sizes()
        _with_env(env, run)
        np.testing.assert_array_equal(u64(got.insert_sizes()), ref.insert_sizes())
        assert got.adapters_read1() == ref.adapters_read1()
        assert got.adapters_read2() == ref.adapters_read2()
        assert got.number_of_adapters_read1 == ref.number_of_adapters_read1
        assert got.number_of_adapters_read2 == ref.number_of_adapters_read2
        assert got.total_reads == ref.total_reads


def test_device_generator_matches_host_generator():
    """the GPU-resident synthetic batches hold exactly the host generator's bytes"""
    from sequali_amd import QCMetrics, synth
    for kind, n in ((synth.ILLUMINA, 3000), (synth.ILLUMINA_R2, 1000), (synth.NANOPORE, 40)):
        host = synth.host_array(kind, 17, n)
        dev = synth.device_array(kind, 17, n)
        assert len(dev) == n
        a, b = QCMetrics(), QCMetrics()
        a.add_record_array(host)
        b.add_record_array(dev)
        assert a.base_count_table() == b.base_count_table()
        assert a.phred_count_table() == b.phred_count_table()
        np.testing.assert_array_equal(host.accumulated_error_rates().view(np.uint64),
                                      dev.accumulated_error_rates().view(np.uint64))


def test_ragged_batch_in_sorted_order_fused():
    """>= 4096 ragged records: processed in length-sorted order; with PerTileQuality
    in the pass, in tile-sorted order"""
    from sequali_amd import AdapterCounter, FusedPass, PerTileQuality, QCMetrics
    rng = np.random.default_rng(77)
    probes = ADAPTER_SETS[0]
    buf, metas, arr = random_batch(rng, 9000, 700, alphabet=b"ACGTNacgt", splice=probes)
    rq, ra, rp = oracle.QCMetrics(), oracle.AdapterCounter(probes), oracle.PerTileQuality()
    rq.add(buf, metas)
    ra.add(buf, metas)
    rp.add(buf, metas)
    for with_pt in (False, True):
        gq, ga, gp = QCMetrics(), AdapterCounter(probes), PerTileQuality()
        FusedPass(gq, ga, gp if with_pt else None).add_record_array(arr)
        compare_qc(rq, gq, metas, arr)
        for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(f), fr)
            np.testing.assert_array_equal(u64(r), rr)
        if with_pt:
            assert gp.number_of_reads == rp.number_of_reads
            for (t, e, c), (tr, er, cr) in zip(gp.get_tile_counts(), rp.get_tile_counts()):
                assert t == tr
                np.testing.assert_allclose(np.array(e), er, rtol=1e-6)
                np.testing.assert_array_equal(u64(c), cr)


def test_nanopore_long_reads_all_modules():
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, synth
    arr = synth.host_array(synth.NANOPORE, 0, 200)
    probes = list(synth.NANOPORE_PROBES)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    metas = arr._metas.copy()
    rq.add(arr.obj, metas)
    ra.add(arr.obj, metas)
    gq, ga = QCMetrics(), AdapterCounter(probes)
    FusedPass(gq, ga).add_record_array(arr)
    compare_qc(rq, gq, metas, arr)
    for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
        np.testing.assert_array_equal(u64(f), fr)
        np.testing.assert_array_equal(u64(r), rr)


def _with_env(env, fn):
    from tests.helpers import with_env
    return with_env(env, fn)


def test_two_million_reads_all_tables_equal_oracle():
    """bench-shaped input at a size the oracle still finishes in seconds: 2 M x 150 bp
    generated in HBM through every kernel that takes such a batch: the fused launch (k_span),
    the same forced through k_ring, k_pass and k_wide, and QCMetrics alone (k_span, k_ring)"""
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, synth
    n = 2_000_000
    dev = synth.device_array(synth.ILLUMINA, 12345, n)
    buf, metas = dev._batch.download()
    probes = list(synth.ILLUMINA_PROBES)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    gq, ga, gq2 = QCMetrics(), AdapterCounter(probes), QCMetrics()
    FusedPass(gq, ga).add_record_array(dev)
    errs = dev.accumulated_error_rates()
    gq2.add_record_array(dev)
    forced = []
    gq3 = QCMetrics()   # QCMetrics alone through k_ring
    _with_env({"SQ_SPAN": "0"}, lambda: (gq3.add_record_array(dev), gq3.flush()))
    for env in ({"SQ_RING": "1"}, {"SQ_NO_WIDE": "1"}, {"SQ_SPAN": "0"}):
        q, a = QCMetrics(), AdapterCounter(probes)
        _with_env(env, lambda: (FusedPass(q, a).add_record_array(dev), q.flush()))
        forced.append((q, a))
    for g in (ga,) + tuple(f[1] for f in forced):
        for (_, f, r), (_, fr, rr) in zip(g.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(f), fr)
            np.testing.assert_array_equal(u64(r), rr)
    for g in (gq, gq2, gq3) + tuple(f[0] for f in forced):
        np.testing.assert_array_equal(u64(g.base_count_table()), rq.base_count_table())
        np.testing.assert_array_equal(u64(g.phred_count_table()), rq.phred_count_table())
        np.testing.assert_array_equal(u64(g.end_anchored_base_count_table()), rq.end_anchored_base_count_table())
        np.testing.assert_array_equal(u64(g.end_anchored_phred_count_table()), rq.end_anchored_phred_count_table())
        np.testing.assert_array_equal(u64(g.gc_content()), rq.gc_content())
        np.testing.assert_array_equal(u64(g.phred_scores()), rq.phred_scores())
    np.testing.assert_array_equal(errs.view(np.uint64), metas["accumulated_error_rate"].view(np.uint64))
    np.testing.assert_array_equal(dev.accumulated_error_rates().view(np.uint64),
                                  metas["accumulated_error_rate"].view(np.uint64))
    assert sum(int(f.sum()) for _, f, _ in ra.get_counts()) > 50_000


def _route_of(run):
    """the kernels a call launched (sq_last_route): a build that does not fit makes a dispatcher fall back to another
    kernel without a word, and a sweep that thinks it covers k_span<7> would be covering k_wide"""
    from sequali_amd._lib import context, lib
    lib().sq_route_reset(context())
    run()
    return (lib().sq_last_route(context()) or b"").decode()


def _uniform_route(U, with_adapters):
    """the first kernel the default dispatch launches for a batch of one read length U"""
    nw = (U + 31) // 32
    if U > 256:
        return None      # the round-1 kernels, by what fits their LDS
    if with_adapters:     # (225-256 bases with adapters: k_wide until round 5; up to 64 bases: k_wide since round 5)
        return f"k_span<{nw},AD,uniform,split>" if nw > 2 else "k_wide<AD>"
    return f"k_span<{nw},QC,uniform,{'split' if nw >= 6 else 'both'}>"


@pytest.mark.parametrize("U", [1, 3, 4, 5, 27, 31, 32, 33, 63, 64, 65, 97, 150, 151, 161, 176, 192, 193, 200, 224, 225, 250, 251, 256, 512])
def test_uniform_length_kernels_every_alignment(U):
    """Batches of one read length have four kernels.  k_span streams 16 records at a time
    through LDS and counts them with four lanes per read, k_ring cuts a read into 32-byte aligned
    windows and rotates them back in registers, k_wide stages 64 positions at a time with four
    lanes per row and lets padding absorb the end of the reads, k_pass is the general one: every
    length class and every start alignment (names of rotating length, so sequence and quality
    starts walk through all residues mod 64), full groups plus a remainder, with and without
    the automaton in the pass (one probe holds an N: padding must not look like it)"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(1000 + U)
    n = 64 * 5 + 37
    probes = ["ACGTACGTACGT"[:min(U, 12)], "GGGGG"[:min(U, 5)], "TTNAC"[:min(U, 5)]]
    names, seqs, quals = [], [], []
    for i in range(n):
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=U, p=[.24, .24, .24, .24, .04]).tobytes().decode()
        if U >= 12 and i % 3 == 0:
            at = int(rng.integers(0, U - 11))
            s = s[:at] + "ACGTACGTACGT" + s[at + 12:]
        if U >= 5 and i % 7 == 0:
            s = s[:U - 4] + "TTNA"  # the N probe one base short of matching at the very end
        names.append("r" * (1 + i % 67))
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    cases = [(False, {}),                                        # QCMetrics alone: k_span (records through LDS by LDS-DMA; k_ring from 257 positions on)
             (False, {"SQ_SPAN_SPLIT_QC": "0"}),                 # QCMetrics alone: k_span, one wave for both streams (the default up to 160 positions)
             (False, {"SQ_SPAN_SPLIT_QC": "1"}),                 # ... a wave per stream (the default from 161 on)
             (False, {"SQ_SPAN": "0"}),                          # QCMetrics alone: k_ring
             (False, {"SQ_NO_RING": "1"}),                       # QCMetrics alone: k_pass
             (True, {}),                                         # + AdapterCounter: k_span (k_wide from 161 positions on)
             (True, {"SQ_SPAN": "0"}),                           # k_wide
             (True, {"SQ_RING": "1"}),                           # k_ring
             (True, {"SQ_NO_WIDE": "1"})]                        # k_pass
    for with_adapters, env in cases:
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(probes)

        def run():
            if with_adapters:
                FusedPass(gq, ga).add_record_array(arr)
            else:
                gq.add_record_array(arr)
            gq.flush()
        route = _route_of(lambda: _with_env(env, run))
        if not env and _uniform_route(U, with_adapters):   # the default dispatch: the kernel DESIGN 4.1b names for this length, not a silent fallback
            assert route.split("+")[0] == _uniform_route(U, with_adapters), (U, with_adapters, route)
        compare_qc(rq, gq, metas, arr)
        if with_adapters:
            for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
                np.testing.assert_array_equal(u64(f), fr)
                np.testing.assert_array_equal(u64(r), rr)


_SWEEP_U = sorted(set(range(1, 65)) | {32 * k + d for k in range(2, 9) for d in (-2, -1, 0, 1, 2) if 32 * k + d <= 256})


@pytest.mark.parametrize("U", _SWEEP_U)
def test_k_span_every_length_up_to_64_and_around_every_window_border(U):
    """k_span<NW, uniform> on EVERY read length from 1 to 64 and two either side of every multiple of 32 up to 256,
    QCMetrics alone and with the adapters (SQ_SPAN_SHORT=1: up to 64 bases the default with adapters is k_wide).  The
    padding masks of a batch of one read length are made from an opaque copy of U (sq_span_kernel.h, `keep_u0`): written
    with the kernel's own U, hipcc (ROCm 7.2) dropped the guards of the f64 chains' last steps and every read of 1-15 and
    33-47 bases summed the text behind its qualities (round 5) -- a sweep over a handful of lengths does not pin that."""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(7000 + U)
    n = 64 * 2 + 21
    probes = ["ACGTACGTACGT"[:min(U, 12)], "GGGGG"[:min(U, 5)]]
    names, seqs, quals = [], [], []
    for i in range(n):
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=U, p=[.24, .24, .24, .24, .04]).tobytes().decode()
        if U >= 12 and i % 3 == 0:
            at = int(rng.integers(0, U - 11)) if i % 2 else U - 12
            s = s[:at] + "ACGTACGTACGT" + s[at + 12:]
        names.append("r" * (1 + i % 67))
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    nw = (U + 31) // 32
    for with_adapters in (False, True):
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(probes)

        def run():
            (FusedPass(gq, ga) if with_adapters else gq).add_record_array(arr)
            gq.flush()
        route = _route_of(lambda: _with_env({"SQ_SPAN_SHORT": "1"}, run))
        want = f"k_span<{nw},AD,uniform,split>" if with_adapters else f"k_span<{nw},QC,uniform,{'split' if nw >= 6 else 'both'}>"
        assert route.split("+")[0] == want, (U, with_adapters, route)
        compare_qc(rq, gq, metas, arr)
        if with_adapters:
            for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
                np.testing.assert_array_equal(u64(f), fr)
                np.testing.assert_array_equal(u64(r), rr)


@pytest.mark.parametrize("max_len,with_adapters,ea", [(160, True, 100), (151, True, 20), (33, True, 100),
                                                      (256, False, 100), (97, False, 300), (12, True, 5),
                                                      (224, True, 100), (256, True, 100), (200, False, 40)])
def test_sorted_spans_every_length(max_len, with_adapters, ea):
    """reads of many lengths through k_span (sorted by length, spans of 16 reads of one length;
    SQ_SPAN_SORTED=1 takes the path at this size): every length from 1 to max_len with a number
    of reads that is no multiple of 16 (the last span of a length is filled up with padding rows),
    several lengths with a single read, adapters at the very end of short and long reads, a probe
    with an N, an end-anchor length shorter and longer than the reads; against the general k_pass
    (SQ_SPAN=0) too"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(4000 + max_len)
    probes = ["ACGTACGTACGT"[:min(max_len, 12)], "GGGGG", "TTNAC"]
    names, seqs, quals = [], [], []
    lengths = list(range(1, max_len + 1)) * 3 + list(rng.integers(1, max_len + 1, size=700)) + [max_len] * 37
    rng.shuffle(lengths)
    for i, L in enumerate(lengths):
        L = int(L)
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[.24, .24, .24, .24, .04]).tobytes().decode()
        if L >= 12 and i % 3 == 0:
            at = int(rng.integers(0, L - 11)) if i % 2 else L - 12
            s = s[:at] + "ACGTACGTACGT" + s[at + 12:]
        if L >= 5 and i % 7 == 0:
            s = s[:L - 5] + ("GGGGG" if i % 2 else "TTNAC")
        names.append("r" * (1 + i % 67))
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(ea), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    # the last: the launches of the window counts side by side on streams of their own
    for env in ({"SQ_SPAN_SORTED": "1"}, {"SQ_SPAN": "0"}):
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(ea), AdapterCounter(probes)

        def run():
            if with_adapters:
                FusedPass(gq, ga).add_record_array(arr)
            else:
                gq.add_record_array(arr)
            gq.flush()
        route = _route_of(lambda: _with_env(env, run))
        if "SQ_SPAN_SORTED" in env:   # every window count of the batch went through a build of k_span<NW,.,sorted,.>
            kinds = {p.split(",")[0] for p in route.split("+") if p.startswith("k_span<")}
            assert kinds == {f"k_span<{nw}" for nw in range(1, (max_len + 31) // 32 + 1)}, route
            assert all(",sorted," in p for p in route.split("+") if p.startswith("k_span<")), route
        compare_qc(rq, gq, metas, arr)
        assert gq.number_of_reads == len(lengths) and gq.max_length == max_len
        if with_adapters:
            for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
                np.testing.assert_array_equal(u64(f), fr)
                np.testing.assert_array_equal(u64(r), rr)


def test_batches_know_their_length_counts():
    """sq_batch_length_counts (what k_span_scatter orders a ragged batch with) for every way a batch is made:
    generated on the device and trimmed, uploaded from host arrays, split on the device, a sealed feeder block"""
    import ctypes as C
    import io
    from sequali_amd import FastqParser, QCMetrics, _lib, synth
    from sequali_amd._lib import context, lib
    from sequali_amd._qc import FastqRecordArrayView, _DeviceBatch

    def counts_of(handle):
        out = (C.c_uint32 * 257)()
        assert lib().sq_batch_length_counts(handle, out) == 1
        return np.array(out[:], dtype=np.int64)

    def want(metas):
        return np.bincount(np.minimum(metas["sequence_length"].astype(np.int64), 256), minlength=257)

    dev = synth.device_array(synth.ILLUMINA, 5, 30_000)
    np.testing.assert_array_equal(counts_of(dev._batch.handle), want(dev._batch.download()[1]))
    _lib.check(lib().sq_synth_trim(dev._batch.handle, 3, 20))
    buf, metas = dev._batch.download()
    assert len(set(metas["sequence_length"].tolist())) > 100
    np.testing.assert_array_equal(counts_of(dev._batch.handle), want(metas))
    long_reads = synth.device_array(synth.NANOPORE, 0, 300)
    np.testing.assert_array_equal(counts_of(long_reads._batch.handle), want(long_reads._batch.download()[1]))
    text, hm = synth.host_records(synth.ILLUMINA, 0, 5000)
    uploaded = _DeviceBatch(lib().sq_batch_upload(context(), text, len(text), hm.ctypes.data, len(hm)))
    np.testing.assert_array_equal(counts_of(uploaded.handle), want(hm))
    consumed = C.c_size_t(0)
    split = _DeviceBatch(lib().sq_batch_from_fastq(context(), text, len(text), C.byref(consumed)))
    np.testing.assert_array_equal(counts_of(split.handle), want(hm))
    # through the default parser: arrays of a feeder block, counted by a module (the block is sealed and uploaded at the flush)
    ragged = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"A" * (1 + i % 97), b"I" * (1 + i % 97)) for i in range(4000))
    q = QCMetrics()
    arrays = list(FastqParser(io.BytesIO(ragged), 4096))
    for a in arrays:
        q.add_record_array(a)
    q.flush()
    assert q.number_of_reads == 4000
    blk = arrays[0]._blk
    np.testing.assert_array_equal(counts_of(blk.array._batch.handle), np.bincount(1 + np.arange(4000) % 97, minlength=257))


def test_sorted_spans_two_million_trimmed_reads():
    """the ragged variant of the bench workload at a size the oracle finishes in seconds: 2 M
    device-generated 150 bp records cut to 50..150 bases, QCMetrics + AdapterCounter; the default
    path at this size is k_span over the sorted reads; then a second batch into the same objects
    through the general k_pass"""
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth
    n = 2_000_000
    dev = synth.device_array(synth.ILLUMINA, 777, n)
    _lib.check(_lib.lib().sq_synth_trim(dev._batch.handle, 99, 50))
    buf, metas = dev._batch.download()
    probes = list(synth.ILLUMINA_PROBES)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    gq, ga = QCMetrics(), AdapterCounter(probes)
    f = FusedPass(gq, ga)
    # rows in order by the batch's length counts (k_span_scatter) / by a radix sort of keys / the general k_pass / the four launches side by side
    for env in ({}, {"SQ_SPAN": "0"}):
        rq.add(buf, metas)
        ra.add(buf, metas)
        _with_env(env, lambda: (f.add_record_array(dev), gq.flush()))
        for (_, fw, rv), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(fw), fr)
            np.testing.assert_array_equal(u64(rv), rr)
        np.testing.assert_array_equal(u64(gq.base_count_table()), rq.base_count_table())
        np.testing.assert_array_equal(u64(gq.phred_count_table()), rq.phred_count_table())
        np.testing.assert_array_equal(u64(gq.end_anchored_base_count_table()), rq.end_anchored_base_count_table())
        np.testing.assert_array_equal(u64(gq.end_anchored_phred_count_table()), rq.end_anchored_phred_count_table())
        np.testing.assert_array_equal(u64(gq.gc_content()), rq.gc_content())
        np.testing.assert_array_equal(u64(gq.phred_scores()), rq.phred_scores())
        np.testing.assert_array_equal(dev.accumulated_error_rates().view(np.uint64),
                                      metas["accumulated_error_rate"].view(np.uint64))
    assert sum(int(fw.sum()) for _, fw, _ in ra.get_counts()) > 50_000


def test_uniform_length_many_adapters_counted_in_device_tables():
    """with more adapters than the per-workgroup LDS hit table takes (8 KB), k_wide counts hits
    straight into the device tables; 150 bp and 250 bp (the longest k_wide's LDS still fits)"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ACGT", np.uint8)
    probes = [rng.choice(letters, size=int(rng.integers(6, 20))).tobytes().decode() for _ in range(20)]
    for U in (150, 250, 151):
        if U == 151:  # more than 64 adapters: two automatons, the second one gets a pass of its own
            probes = probes + [rng.choice(letters, size=int(rng.integers(5, 12))).tobytes().decode() for _ in range(50)]
        n = 64 * 9 + 5
        names, seqs, quals = [], [], []
        for i in range(n):
            s = rng.choice(letters, size=U).tobytes().decode()
            for _ in range(int(rng.integers(0, 3))):
                w = probes[int(rng.integers(0, len(probes)))]
                at = int(rng.integers(0, U - len(w) + 1))
                s = s[:at] + w + s[at + len(w):]
            names.append(f"q{i}")
            seqs.append(s)
            quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
        buf, metas = oracle.make_batch(names, seqs, quals)
        rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
        rq.add(buf, metas)
        ra.add(buf, metas)
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(probes)
        FusedPass(gq, ga).add_record_array(arr)
        gq.flush()
        compare_qc(rq, gq, metas, arr)
        total = 0
        for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(f), fr)
            np.testing.assert_array_equal(u64(r), rr)
            total += int(fr.sum())
        assert total > n // 2


def test_nanostats_after_qcmetrics_on_device_batches():
    """NanoStats reads accumulated_error_rate where QCMetrics left it in HBM: synthetic
    nanopore reads generated on the device, two batches, nothing on the host in between"""
    from sequali_amd import NanoStats, QCMetrics, synth
    ref_q, ref_n = oracle.QCMetrics(), oracle.NanoStats()
    got_q, got_n = QCMetrics(), NanoStats()
    for first, n in ((0, 3000), (3000, 1500)):
        buf, metas = synth.host_records(synth.NANOPORE, first, n)
        ref_q.add(buf, metas)
        ref_n.add(buf, metas)
        dev = synth.device_array(synth.NANOPORE, first, n)
        got_q.add_record_array(dev)
        got_n.add_record_array(dev)
    assert got_n.number_of_reads == ref_n.number_of_reads == 4500
    assert got_n.skipped_reason is None and not ref_n.skipped
    assert (got_n.minimum_time, got_n.maximum_time) == (ref_n.minimum_time, ref_n.maximum_time)
    got, want = got_n.nano_infos(), ref_n.nano_infos()
    for f in ("start_time", "channel_id", "length", "parent_id_hash"):
        np.testing.assert_array_equal(got[f], want[f])
    np.testing.assert_array_equal(got["cumulative_error_rate"].view(np.uint64),
                                  want["cumulative_error_rate"].view(np.uint64))
    assert float(want["cumulative_error_rate"].sum()) > 0


def test_nanostats_minimum_time_restarts_after_a_zero_timestamp():
    """records without an st tag have start_time 0, which resets minimum_time (:5319-5321)"""
    import struct
    from sequali_amd import FastqRecordArrayView, NanoStats
    def st(ts):
        return b"stZ" + ts + b"\x00"
    ch = b"chC\x07"
    tags = [ch + st(b"2021-09-30T11:34:08Z"), ch + st(b"2020-01-01T00:00:00Z"), ch,
            ch + st(b"2022-02-02T02:02:02Z"), ch + st(b"2021-12-12T12:12:12Z"), ch,
            ch + st(b"2023-03-03T03:03:03Z")]
    names = [f"r{i}" for i in range(len(tags))]
    buf, metas = oracle.make_view_batch(names, ["ACGT"] * len(tags), ["IIII"] * len(tags), tags)
    for cut in (len(tags), 3, 5, 6):   # also with the zero as the last record of a batch
        ref, got = oracle.NanoStats(), NanoStats()
        for a, b in ((0, cut), (cut, len(tags))):
            if b > a:
                ref.add(buf, metas[a:b].copy())
                got.add_record_array(FastqRecordArrayView._from_buffer(buf, metas[a:b].copy()))
            assert (got.minimum_time, got.maximum_time) == (ref.minimum_time, ref.maximum_time)
        assert got.number_of_reads == ref.number_of_reads == len(tags)


@pytest.mark.parametrize("which", [0, 2, 3])
def test_long_reads_in_segments(which):
    """>= 4096 reads longer than 512 bases take the segment kernels (k_read_sums, k_seg,
    k_adapter_first): patterns planted across the 256-position segment borders (the automaton
    is restarted 64 positions in front of a segment), repeated patterns (only the first
    occurrence in a read counts, whichever segment sees it), up to 64 characters long,
    more than 64 patterns (a second automaton in its own pass)"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(300 + which)
    adapters = ADAPTER_SETS[which]
    n = 4300
    names, seqs, quals = [], [], []
    for i in range(n):
        L = int(rng.integers(0, 2600)) if i % 7 else int(rng.integers(0, 40))
        s = bytearray(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[.245, .245, .245, .245, .02]).tobytes())
        for _ in range(int(rng.integers(0, 4))):
            word = adapters[int(rng.integers(0, len(adapters)))].encode()
            if len(word) <= L:
                border = 256 * int(rng.integers(0, L // 256 + 1))
                at = min(max(border - int(rng.integers(0, len(word) + 2)), 0), L - len(word))
                s[at:at + len(word)] = word
        names.append(f"r{i}")
        seqs.append(s.decode())
        quals.append((rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes().decode())
    buf, metas = oracle.make_batch(names, seqs, quals)
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(adapters)
    rq.add(buf, metas)
    ra.add(buf, metas)
    # k_span<LONG> where the adapters allow it (<= 13 characters); the same with the reads walked in blocks (all segments of 512 / 4096 reads before the next ones); k_seg
    for env in ({}, {"SQ_LONG": "0"}):
        arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
        gq, ga = QCMetrics(), AdapterCounter(adapters)
        _with_env(env, lambda: (FusedPass(gq, ga).add_record_array(arr), gq.flush()))
        compare_qc(rq, gq, metas, arr)
        for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(f), fr)
            np.testing.assert_array_equal(u64(r), rr)
    assert sum(int(f.sum()) for _, f, _ in ra.get_counts()) > 500
    gq2 = QCMetrics()   # QCMetrics alone: k_seg without the automaton
    arr2 = FastqRecordArrayView._from_buffer(buf, metas.copy())
    gq2.add_record_array(arr2)
    compare_qc(rq, gq2, metas, arr2)


def _pertile_equal(got, ref):
    assert got.number_of_reads == ref.number_of_reads
    assert got.max_length == ref.max_length
    gt, rt = got.get_tile_counts(), ref.get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (t, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_array_equal(u64(c), cr)
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6, atol=0)   # f64 sums, other order


def _tile_batch(rng, n, U, tile_of, bad_at=None):
    """n reads of length U, Illumina names with rotating lengths (sequence and quality starts walk
    through every alignment), tile_of(i) the tile of read i, optionally one header without a tile"""
    names, seqs, quals = [], [], []
    for i in range(n):
        name = f"M{'x' * (i % 61)}:1:F:{i % 4}:{tile_of(i)}:{i}:7 1:N:0:X"
        if bad_at is not None and i == bad_at:
            name = f"read{i} no tile here"
        names.append(name)
        seqs.append(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=U).tobytes().decode())
        quals.append((rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode())
    return oracle.make_batch(names, seqs, quals)


@pytest.mark.parametrize("U", [1, 3, 4, 5, 27, 31, 32, 33, 63, 64, 65, 97, 150, 151, 251, 512])
def test_ptq_kernel_every_length_and_alignment(U):
    """PerTileQuality alone on a batch of one read length is k_ptspan up to 256 positions (the
    batch streamed as stored, the workgroup's whole table in LDS) and k_ptq beyond or with
    SQ_SPAN=0 (_qcmodule.c:3124-3222 on the qualities only, 64 bytes per row and visit, per-wave
    sums of the tile the groups are in):
    the 16 length classes x rotating alignments with (a) random tiles in >= 4096 reads (the
    tile-sorted walk), (b) tiles in runs with a seam inside a group of 64 and a trailing partial
    group (walked as stored), (c) a header without a tile in the middle of the batch (the module
    stops there for good, :3137-3148).  Every case also through the general kernel
    (SQ_NO_PTQ=1) and against the oracle."""
    from sequali_amd import FastqRecordArrayView, PerTileQuality
    rng = np.random.default_rng(5000 + U)
    tiles = [1101, 1102, 2205, 7, 99239, 1213]
    cases = [
        (64 * 66 + 37, lambda i: tiles[int(rng.integers(0, len(tiles)))], None),     # (a)
        (64 * 9 + 21, lambda i: tiles[min(i // 150, len(tiles) - 1)], None),          # (b) seams at 150, 300, ...
        (64 * 6 + 5, lambda i: tiles[(i // 97) % len(tiles)], 64 * 3 + 17),           # (c)
    ]
    for n, tile_of, bad_at in cases:
        buf, metas = _tile_batch(rng, n, U, tile_of, bad_at)
        ref = oracle.PerTileQuality()
        ref.add(buf, metas)
        for env in ({}, {"SQ_SPAN": "0"}, {"SQ_NO_PTQ": "1"}):
            got = PerTileQuality()
            arr = FastqRecordArrayView._from_buffer(buf, metas.copy())
            _with_env(env, lambda: got.add_record_array(arr))
            _pertile_equal(got, ref)
            if bad_at is not None:
                assert ref.skipped and ref.skipped_record == bad_at
                assert got.skipped_reason == f"Can not parse header: 'read{bad_at} no tile here'"
                # the module stays off: a later batch changes nothing
                arr2 = FastqRecordArrayView._from_buffer(buf, metas.copy())
                _with_env(env, lambda: got.add_record_array(arr2))
                _pertile_equal(got, ref)


def test_config4_nanopore_reads_through_the_segment_kernels():
    """BASELINE config 4 at a size the oracle still finishes in seconds: 4500 reads of the
    synthetic nanopore generator (about 43 M bases, lengths 200 .. 74489: one read longer than
    64 kb, two blocks of 4096 reads) through FusedPass(QCMetrics, AdapterCounter(14 probes)),
    which is k_read_sums + k_seg + k_adapter_first at that size: every table, the first
    occurrences of the adapters and the bits of accumulated_error_rate against the oracle;
    the stripes route (SQ_NO_SEGMENTS=1) as the cross-check"""
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, synth
    first, n = 4562, 4500
    probes = list(synth.NANOPORE_PROBES)
    assert len(probes) == 14
    buf, metas = synth.host_records(synth.NANOPORE, first, n)
    assert metas["sequence_length"].max() > 65536 and n > 4096
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    for env in ({}, {"SQ_LONG": "0"}, {"SQ_NO_SEGMENTS": "1"}):   # k_span<LONG>; k_seg; stripes of k_pass
        dev = synth.device_array(synth.NANOPORE, first, n)
        gq, ga = QCMetrics(), AdapterCounter(probes)
        _with_env(env, lambda: (FusedPass(gq, ga).add_record_array(dev), gq.flush()))
        assert gq.number_of_reads == rq.number_of_reads and gq.max_length == rq.max_length
        np.testing.assert_array_equal(u64(gq.base_count_table()), rq.base_count_table())
        np.testing.assert_array_equal(u64(gq.phred_count_table()), rq.phred_count_table())
        np.testing.assert_array_equal(u64(gq.end_anchored_base_count_table()), rq.end_anchored_base_count_table())
        np.testing.assert_array_equal(u64(gq.end_anchored_phred_count_table()), rq.end_anchored_phred_count_table())
        np.testing.assert_array_equal(u64(gq.gc_content()), rq.gc_content())
        np.testing.assert_array_equal(u64(gq.phred_scores()), rq.phred_scores())
        np.testing.assert_array_equal(dev.accumulated_error_rates().view(np.uint64),
                                      metas["accumulated_error_rate"].view(np.uint64))
        for (_, f, r), (_, fr, rr) in zip(ga.get_counts(), ra.get_counts()):
            np.testing.assert_array_equal(u64(f), fr)
            np.testing.assert_array_equal(u64(r), rr)
    assert sum(int(f.sum()) for _, f, _ in ra.get_counts()) > 10   # chance matches only: the generator plants no probes


@pytest.mark.parametrize("by_tile", [True, False])
def test_config3_one_million_pairs(by_tile):
    """BASELINE config 3 at a size the oracle finishes in seconds: 1 M device-generated pairs (96 tiles: in the order a
    sequencer writes, 65536 reads of a tile in a row, or a random tile per read), (QCMetrics + PerTileQuality) x 2
    through FusedPass, InsertSizeMetrics and a paired DedupEstimator: every getter against the oracle.  By tile:
    k_span<QC+PT> (PerTileQuality rides in QCMetrics' pass, sq_pair.hip) + k_pt_fold per side; random tiles: the same pass
    for the tile ids and k_ptspan (the 116 KB tile table that leaves it 7 waves) for the table; k_isz_span +
    k_isz_adapters, k_dedup_hash -- and once more with the passes of round 2 (SQ_PT_FUSED=0: k_tile_parse, k_span,
    k_ptspan) and with SQ_SPAN=0 (the round-1 kernels) as the cross-checks.
    Reference: _qcmodule.c:3088-3121, :3189-3220, :5668-5707, :4488-4517."""
    from sequali_amd import DedupEstimator, FusedPass, InsertSizeMetrics, PerTileQuality, QCMetrics, synth
    n, first = 1_000_000, 7_000_000
    d1 = synth.device_array(synth.ILLUMINA_BY_TILE if by_tile else synth.ILLUMINA, first, n)
    d2 = synth.device_array(synth.ILLUMINA_R2_BY_TILE if by_tile else synth.ILLUMINA_R2, first, n)
    b1, m1 = d1._batch.download()
    b2, m2 = d2._batch.download()
    rq1, rq2, rp1, rp2 = oracle.QCMetrics(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.PerTileQuality()
    rz, rd = oracle.InsertSizeMetrics(), oracle.DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
    rq1.add(b1, m1); rq2.add(b2, m2); rp1.add(b1, m1); rp2.add(b2, m2)
    rz.add_pair(b1, m1, b2, m2)
    rd.add_pair(b1, m1, b2, m2)
    assert len(rp1.get_tile_counts()) == (96 if not by_tile else len({(i >> 16) % 96 for i in range(first, first + n, 4096)}))

    def run():
        q1, q2, p1, p2 = QCMetrics(), QCMetrics(), PerTileQuality(), PerTileQuality()
        z, d = InsertSizeMetrics(), DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
        FusedPass(q1, None, p1).add_record_array(d1)
        FusedPass(q2, None, p2).add_record_array(d2)
        z.add_record_array_pair(d1, d2)
        d.add_record_array_pair(d1, d2)
        q1.flush(); q2.flush()
        return q1, q2, p1, p2, z, d

    # the default (PerTileQuality rides in QCMetrics' pass) / the passes of round 2 (the pass over the headers on a stream
    # of its own, beside the counting) / that pass on the work stream / the round-1 kernels
    for env in ({}, {"SQ_PT_FUSED": "0"}, {"SQ_SPAN": "0"}):
        route = _route_of(lambda: _with_env(env, run))
        if not env:      # the default since round 5
            want = "k_span<5,QCPT,uniform,both>+k_pt_fold" if by_tile else "k_span<5,QCPT,uniform,both>+k_ptspan<5>"
            assert route.startswith(want + "+" + want + "+k_isz_span<5>"), route
        elif env == {"SQ_PT_FUSED": "0"}:
            assert route.startswith("k_span<5,QC,uniform,both>+k_ptspan<5>+k_span<5,QC,uniform,both>+k_ptspan<5>+k_isz_span<5>"), route
        q1, q2, p1, p2, z, d = _with_env(env, run)
        for g, r, dev, metas in ((q1, rq1, d1, m1), (q2, rq2, d2, m2)):
            compare_qc(r, g, metas, dev)
        for g, r in ((p1, rp1), (p2, rp2)):
            assert g.number_of_reads == r.number_of_reads == n
            gt, rt = g.get_tile_counts(), r.get_tile_counts()
            assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
            for (t, e, c), (_, er, cr) in zip(gt, rt):
                np.testing.assert_allclose(np.array(e), er, rtol=1e-6, err_msg=f"tile {t}")
                np.testing.assert_array_equal(u64(c), cr, err_msg=f"tile {t}")
        assert z.total_reads == rz.total_reads and z.number_of_adapters_read1 == rz.number_of_adapters_read1
        assert z.number_of_adapters_read2 == rz.number_of_adapters_read2
        np.testing.assert_array_equal(u64(z.insert_sizes()), rz.insert_sizes())
        assert z.adapters_read1() == rz.adapters_read1() and z.adapters_read2() == rz.adapters_read2()
        np.testing.assert_array_equal(u64(d.duplication_counts()), rd.duplication_counts())
        assert d._modulo_bits == rd._modulo_bits and d.tracked_sequences == rd.tracked_sequences


@pytest.mark.parametrize("bad_byte", [0x20, 0x80])
def test_long_reads_with_an_invalid_phred_byte(bad_byte):
    """>= 4096 long reads, one of them with a byte that is no phred character (0x80: what BAM
    quality 95 becomes, and the code k_span<LONG> pads with): k_read_sums flags the batch, the
    per-position pass falls back to k_seg, the flush raises the reference's ValueError and leaves
    the reference's state behind it (:2073-2075, :2102-2105)"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics
    rng = np.random.default_rng(bad_byte)
    n, bad = 4200, 2777
    letters = np.frombuffer(b"ACGT", np.uint8)
    lens = rng.integers(520, 1400, size=n)
    parts, metas, pos = [], np.zeros(n, dtype=oracle.META_DTYPE), 0
    for i in range(n):
        L = int(lens[i])
        q = (rng.integers(0, 94, size=L) + 33).astype(np.uint8)
        if i == bad:
            q[L // 3] = bad_byte
        name = b"r%d" % i
        rec = b"@" + name + b"\n" + rng.choice(letters, size=L).tobytes() + b"\n+\n" + q.tobytes() + b"\n"
        metas[i] = (pos + 1, len(name), len(name) + 1, L, len(name) + 1 + L + 3, len(name) + 1 + 2 * L + 3, 0, 0.0)
        parts.append(rec)
        pos += len(rec)
    buf = b"".join(parts)
    ref, ref_metas = oracle.QCMetrics(), metas.copy()
    with pytest.raises(ValueError):
        ref.add(buf, ref_metas)
    probes = ["ACGTACGTACGT", "GGGGGGGGGGGG"]
    m, a = QCMetrics(), AdapterCounter(probes)

    def run():
        FusedPass(m, a).add_record_array(FastqRecordArrayView._from_buffer(buf, metas.copy()))
        with pytest.raises(ValueError, match="Not a valid phred character"):
            m.flush()
    route = _route_of(run)
    assert "k_seg" in route and "k_span<8,AD,long>" not in route, route
    assert m.number_of_reads == ref.number_of_reads and m.max_length == ref.max_length
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        np.testing.assert_array_equal(u64(getattr(m, name)()), getattr(ref, name)(), err_msg=name)
