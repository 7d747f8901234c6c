"""The oracle (oracle/sq_oracle.c, a restatement) against the reference ITSELF (oracle/_ref/_qc.abi3.so: the
reference's own _qcmodule.c compiled where it lies, oracle/Makefile) on random inputs, random module parameters and
random array boundaries -- the golden vectors of tests/golden pin the oracle on fixed inputs, this pins it on
inputs nobody chose.  CPU only; skipped where oracle/_ref has not been built (it is built in the container
that holds /root/reference and travels with the snapshot; the reference's sources do not).

Every getter SURVEY 8a lists, bit for bit: the tables, the floating-point sums (the oracle keeps the reference's
order of additions), the slot order of the hash tables."""
import importlib.util
import io
import os
import sys
import warnings

import numpy as np
import pytest

from oracle import oracle

REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _reference():
    path = os.path.join(REF_DIR, "_qc.abi3.so")
    if not os.path.exists(path):
        return None
    if "_qc" in sys.modules:
        return sys.modules["_qc"]
    spec = importlib.util.spec_from_file_location("_qc", path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError:
        return None
    sys.modules["_qc"] = mod
    return mod


REF = _reference()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")

ADAPTER_SETS = [["AGATCGGAAGAG", "CTGTCTCTTATA", "GGGGGGGGGGGG"], ["ACG", "NN", "GTAC", "TTTTTTTT"],
                ["ACGT" * 16, "A" * 40, "TGGAATTCTCGG", "GATCGTCGGACT", "AAAAAAAAAAAA", "CCCCCC"]]


def draw(rng, n, max_len, uniform, names_kind, adapters, pool=None):
    """n records: (names, seqs, quals).  A quarter of the reads repeat one of 50 earlier ones (duplicates for the
    estimator and the fragment table), some carry an adapter, some bases are lower case or N"""
    names, seqs, quals = [], [], []
    U = int(rng.integers(1, max_len + 1))
    pool = [] if pool is None else pool
    for i in range(n):
        L = U if uniform else int(rng.integers(0, max_len + 1))
        if pool and rng.random() < 0.25:
            s = pool[int(rng.integers(0, len(pool)))]
            s = (s * (L // max(len(s), 1) + 1))[:L]
        else:
            s = rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), size=L,
                           p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).tobytes().decode()
            if len(pool) < 50:
                pool.append(s)
        if adapters and rng.random() < 0.3:
            w = adapters[int(rng.integers(0, len(adapters)))]
            if len(w) <= L:
                at = int(rng.integers(0, L - len(w) + 1))
                s = s[:at] + w + s[at + len(w):]
        L = len(s)      # (an empty read in the pool repeats to an empty read)
        seqs.append(s)
        quals.append((rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes().decode())
        tile = int(rng.choice([1101, 1102, 2205, 7, 99239, 0]))
        if names_kind == "illumina":
            names.append(f"M{'x' * (i % 9)}:1:F:{i % 4}:{tile}:{i}:{L} 1:N:0:X")
        elif names_kind == "breaks":       # a header PerTileQuality cannot parse somewhere in the middle (:3137-3148)
            names.append(f"M:1:F:{i % 4}:{tile}:{i}:{L}" if i != n // 2 else f"read{i} no tile")
        else:
            names.append(f"read{i} ch={i % 512}")
    return names, seqs, quals


def fastq(names, seqs, quals) -> bytes:
    return "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in zip(names, seqs, quals)).encode()


def reference_arrays(text: bytes, buffer_size: int):
    return list(REF.FastqParser(io.BytesIO(text), buffer_size))


def slices_like(arrays):
    """the record ranges the reference's parser cut the file into"""
    out, at = [], 0
    for a in arrays:
        out.append((at, at + len(a)))
        at += len(a)
    return out


def u64(a):
    return np.array(a, dtype=np.uint64)


@pytest.mark.parametrize("seed", range(60))
def test_single_end_modules(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 17, 64, 400, 1500]))
    max_len = int(rng.choice([5, 40, 151, 300, 1200]))
    adapters = ADAPTER_SETS[int(rng.integers(0, len(ADAPTER_SETS)))]
    kind = str(rng.choice(["illumina", "illumina", "breaks", "plain"]))
    names, seqs, quals = draw(rng, n, max_len, bool(rng.random() < 0.4), kind, adapters)
    text = fastq(names, seqs, quals)
    arrays = reference_arrays(text, int(rng.choice([64, 4096, 1 << 20])))     # small buffers grow: many arrays
    buf, metas = oracle.make_batch(names, seqs, quals)
    ea = int(rng.choice([0, 7, 100, 300]))
    okw = dict(max_unique_fragments=int(rng.choice([50, 700, 5_000_000])), sample_every=int(rng.choice([1, 3, 8])),
               fragment_length=int(rng.choice([5, 21, 31])), bases_from_start=int(rng.choice([100, 10, 0])),
               bases_from_end=int(rng.choice([100, 30]))
               )
    dkw = dict(max_stored_fingerprints=int(rng.choice([100, 300, 1_000_000])),
               front_sequence_offset=int(rng.choice([0, 8, 64])), back_sequence_offset=int(rng.choice([0, 8, 64])),
               front_sequence_length=int(rng.choice([8, 3, 16])), back_sequence_length=int(rng.choice([8, 0, 5])))
    ref = dict(q=REF.QCMetrics(ea), a=REF.AdapterCounter(adapters), p=REF.PerTileQuality(),
               o=REF.OverrepresentedSequences(**okw), d=REF.DedupEstimator(**dkw))
    got = dict(q=oracle.QCMetrics(ea), a=oracle.AdapterCounter(adapters), p=oracle.PerTileQuality(),
               o=oracle.OverrepresentedSequences(**okw), d=oracle.DedupEstimator(**dkw))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for arr in arrays:
            for m in ref.values():
                m.add_record_array(arr)
        for lo, hi in slices_like(arrays):      # the same boundaries: the fragment table's staging is per call
            x = metas[lo:hi].copy()
            for m in got.values():
                m.add(buf, x)
            metas[lo:hi] = x
    r, g = ref["q"], got["q"]
    assert (g.number_of_reads, g.max_length) == (r.number_of_reads, r.max_length)
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        np.testing.assert_array_equal(getattr(g, name)(), u64(getattr(r, name)()), err_msg=name)
    r, g = ref["a"], got["a"]
    assert (g.number_of_sequences, g.max_length) == (r.number_of_sequences, r.max_length)
    for (an, f, rv), (rn, fr, rr) in zip(g.get_counts(), r.get_counts()):
        assert an == rn
        np.testing.assert_array_equal(f, u64(fr))
        np.testing.assert_array_equal(rv, u64(rr))
    r, g = ref["p"], got["p"]
    assert (g.number_of_reads, g.max_length) == (r.number_of_reads, r.max_length)
    assert g.skipped == (r.skipped_reason is not None)
    rt = r.get_tile_counts()
    gt = g.get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (_, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_array_equal(np.asarray(e, dtype=np.float64).view(np.uint64), np.array(er, dtype=np.float64).view(np.uint64))
        np.testing.assert_array_equal(c, u64(cr))
    r, g = ref["o"], got["o"]
    for k in ("number_of_sequences", "sampled_sequences", "total_fragments", "collected_unique_fragments"):
        assert getattr(g, k) == getattr(r, k), k
    assert g.sequence_counts() == r.sequence_counts()
    assert [(c, s) for c, _, s in g.overrepresented_sequences()] == [(c, s) for c, _, s in r.overrepresented_sequences()]
    assert [f for _, f, _ in g.overrepresented_sequences()] == [f for _, f, _ in r.overrepresented_sequences()]
    r, g = ref["d"], got["d"]
    assert (g._modulo_bits, g.tracked_sequences, g._hash_table_size) == (r._modulo_bits, r.tracked_sequences, r._hash_table_size)
    np.testing.assert_array_equal(g.duplication_counts(), u64(r.duplication_counts()))     # slot order
    # the error rate QCMetrics writes back into the array (:2126), through the views of the reference's arrays: NanoStats reads it
    ns = REF.NanoStats()
    at = 0
    for arr in arrays:
        ns.add_record_array(arr)
    rates = np.array([info.cumulative_error_rate for info in ns.nano_info_iterator()], dtype=np.float64)
    np.testing.assert_array_equal(rates.view(np.uint64), metas["accumulated_error_rate"].view(np.uint64)[:len(rates)])


@pytest.mark.parametrize("seed", range(40))
def test_paired_modules(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 33, 500, 2000]))
    max_len = int(rng.choice([5, 14, 40, 151, 260]))
    pool = []
    n1, s1, q1 = draw(rng, n, max_len, bool(rng.random() < 0.5), "illumina", None, pool)
    n2, s2, q2 = draw(rng, n, max_len, bool(rng.random() < 0.5), "illumina", None, pool)
    comp = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")
    for i in range(n):     # real pairs for the overlap scan: read 2 = the reverse complement of a piece of read 1 + an adapter
        if rng.random() < 0.5 and len(s1[i]) >= 20:
            ins = int(rng.integers(16, len(s1[i]) + 1))
            frag = s1[i][:ins]
            r2 = frag[::-1].translate(comp) + "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
            L2 = len(s2[i])
            s2[i] = (r2 + s2[i])[:max(L2, 16)] if L2 else ""
            q2[i] = (q2[i] + "I" * len(s2[i]))[:len(s2[i])]
    # The reference's fingerprint store is PyMem_Malloc'ed and never cleared (:4352): a pair shorter than the fingerprint
    # shows whatever the allocator left there until a long pair has rewritten all 16 bytes.  The oracle (and the
    # library) start from zeros; to compare, the first pair is a long one.
    s1[0] = s2[0] = "ACGTTGCAACGTTGCAAC"
    q1[0] = q2[0] = "I" * 18
    a1 = reference_arrays(fastq(n1, s1, q1), 1 << 24)
    a2 = reference_arrays(fastq(n2, s2, q2), 1 << 24)
    assert len(a1) == len(a2) == 1
    b1, m1 = oracle.make_batch(n1, s1, q1)
    b2, m2 = oracle.make_batch(n2, s2, q2)
    dkw = dict(max_stored_fingerprints=int(rng.choice([100, 300, 1_000_000])),
               front_sequence_offset=int(rng.choice([0, 8])), back_sequence_offset=int(rng.choice([0, 8])))
    rd, gd = REF.DedupEstimator(**dkw), oracle.DedupEstimator(**dkw)
    rz, gz = REF.InsertSizeMetrics(), oracle.InsertSizeMetrics()
    cuts = sorted({0, n, *(int(x) for x in rng.integers(0, n + 1, size=2))})
    rd.add_record_array_pair(a1[0], a2[0])
    rz.add_record_array_pair(a1[0], a2[0])
    for lo, hi in zip(cuts[:-1], cuts[1:]):      # the oracle in pieces: its state carries over (the store of short pairs, :4512)
        gd.add_pair(b1, m1[lo:hi].copy(), b2, m2[lo:hi].copy())
        gz.add_pair(b1, m1[lo:hi].copy(), b2, m2[lo:hi].copy())
    assert (gd._modulo_bits, gd.tracked_sequences) == (rd._modulo_bits, rd.tracked_sequences)
    np.testing.assert_array_equal(gd.duplication_counts(), u64(rd.duplication_counts()))
    assert (gz.total_reads, gz.number_of_adapters_read1, gz.number_of_adapters_read2) == \
        (rz.total_reads, rz.number_of_adapters_read1, rz.number_of_adapters_read2)
    np.testing.assert_array_equal(gz.insert_sizes(), u64(rz.insert_sizes()))
    assert gz.adapters_read1() == list(rz.adapters_read1())     # slot order
    assert gz.adapters_read2() == list(rz.adapters_read2())


@pytest.mark.parametrize("seed", range(12))
def test_a_byte_that_is_no_phred_character(seed):
    """:2102-2105 (QCMetrics) and :3212-3215 (PerTileQuality): ValueError with the character; what was counted up to
    there stays counted -- the same message and the same tables afterwards"""
    rng = np.random.default_rng(11000 + seed)
    n = int(rng.choice([1, 40, 300]))
    names, seqs, quals = draw(rng, n, int(rng.choice([30, 151, 400])), bool(seed % 2), "illumina", None)
    victims = [i for i in range(n) if len(seqs[i]) > 0]
    if not victims:
        pytest.skip("no read with a base")
    v = victims[int(rng.integers(0, len(victims)))]
    at = int(rng.integers(0, len(quals[v])))
    bad = chr(int(rng.choice([32, 127, 10 + 22])))          # ' ', DEL, ' ' again: below '!' and above '~'
    quals[v] = quals[v][:at] + bad + quals[v][at + 1:]
    arrays = reference_arrays(fastq(names, seqs, quals), 1 << 22)
    buf, metas = oracle.make_batch(names, seqs, quals)
    for make_ref, make_got in ((lambda: REF.QCMetrics(), lambda: oracle.QCMetrics()),
                               (lambda: REF.PerTileQuality(), lambda: oracle.PerTileQuality())):
        r, g = make_ref(), make_got()
        with pytest.raises(ValueError) as er:
            r.add_record_array(arrays[0])
        with pytest.raises(ValueError) as eg:
            g.add(buf, metas.copy())
        assert str(eg.value) == str(er.value)
        assert (g.number_of_reads, g.max_length) == (r.number_of_reads, r.max_length)
        if isinstance(g, oracle.QCMetrics):
            for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                         "end_anchored_phred_count_table", "gc_content", "phred_scores"):
                np.testing.assert_array_equal(getattr(g, name)(), u64(getattr(r, name)()), err_msg=name)
        else:
            rt, gt = r.get_tile_counts(), g.get_tile_counts()
            assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
            for (_, e, c), (_, er_, cr) in zip(gt, rt):
                np.testing.assert_array_equal(np.asarray(e, dtype=np.float64).view(np.uint64), np.array(er_, dtype=np.float64).view(np.uint64))
                np.testing.assert_array_equal(c, u64(cr))


@pytest.mark.parametrize("kind,first,n", [("ILLUMINA", 33_000_000, 20_000), ("ILLUMINA", 0, 5_000), ("NANOPORE", 777, 250),
                                          ("ILLUMINA_BY_TILE", 1_000_000, 8_000)])
def test_slices_of_the_benchmark_generator(kind, first, n):
    """what bench.py feeds the GPU (sequali_amd/synth.py, counter-based: any slice on its own), far from the slices the golden
    vectors hold: all six single-end modules with caps small enough to be crossed (8 modulo increments of the estimator, the
    fragment table full)"""
    from sequali_amd import synth
    k = getattr(synth, kind)
    buf, metas = synth.host_records(k, first, n)
    text = bytes(buf)
    arrays = reference_arrays(text, 1 << 26)
    assert sum(len(a) for a in arrays) == n
    probes = list(synth.NANOPORE_PROBES if kind == "NANOPORE" else synth.ILLUMINA_PROBES)
    ref = dict(q=REF.QCMetrics(), a=REF.AdapterCounter(probes), p=REF.PerTileQuality(),
               o=REF.OverrepresentedSequences(max_unique_fragments=3000, sample_every=2), d=REF.DedupEstimator(300))
    got = dict(q=oracle.QCMetrics(), a=oracle.AdapterCounter(probes), p=oracle.PerTileQuality(),
               o=oracle.OverrepresentedSequences(max_unique_fragments=3000, sample_every=2), d=oracle.DedupEstimator(300))
    metas = metas.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for arr in arrays:
            for m in ref.values():
                m.add_record_array(arr)
        for m in got.values():
            m.add(buf, metas)
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        np.testing.assert_array_equal(getattr(got["q"], name)(), u64(getattr(ref["q"], name)()), err_msg=name)
    for (_, f, rv), (_, fr, rr) in zip(got["a"].get_counts(), ref["a"].get_counts()):
        np.testing.assert_array_equal(f, u64(fr))
        np.testing.assert_array_equal(rv, u64(rr))
    assert got["p"].skipped == (ref["p"].skipped_reason is not None)
    gt, rt = got["p"].get_tile_counts(), ref["p"].get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (_, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_array_equal(np.asarray(e, dtype=np.float64).view(np.uint64), np.array(er, dtype=np.float64).view(np.uint64))
        np.testing.assert_array_equal(c, u64(cr))
    assert got["o"].sequence_counts() == ref["o"].sequence_counts()
    assert got["o"].collected_unique_fragments == ref["o"].collected_unique_fragments
    assert (got["d"]._modulo_bits, got["d"].tracked_sequences) == (ref["d"]._modulo_bits, ref["d"].tracked_sequences)
    np.testing.assert_array_equal(got["d"].duplication_counts(), u64(ref["d"].duplication_counts()))
    if kind != "NANOPORE":
        assert got["d"]._modulo_bits >= 4


def test_pairs_of_the_benchmark_generator():
    from sequali_amd import synth
    first, n = 12_345_678, 10_000
    b1, m1 = synth.host_records(synth.ILLUMINA, first, n)
    b2, m2 = synth.host_records(synth.ILLUMINA_R2, first, n)
    a1, a2 = reference_arrays(bytes(b1), 1 << 26), reference_arrays(bytes(b2), 1 << 26)
    rd, rz = REF.DedupEstimator(400, front_sequence_offset=0, back_sequence_offset=0), REF.InsertSizeMetrics()
    gd, gz = oracle.DedupEstimator(400, front_sequence_offset=0, back_sequence_offset=0), oracle.InsertSizeMetrics()
    rd.add_record_array_pair(a1[0], a2[0])
    rz.add_record_array_pair(a1[0], a2[0])
    gd.add_pair(b1, m1.copy(), b2, m2.copy())
    gz.add_pair(b1, m1.copy(), b2, m2.copy())
    assert (gd._modulo_bits, gd.tracked_sequences) == (rd._modulo_bits, rd.tracked_sequences)
    np.testing.assert_array_equal(gd.duplication_counts(), u64(rd.duplication_counts()))
    np.testing.assert_array_equal(gz.insert_sizes(), u64(rz.insert_sizes()))
    assert int(gz.insert_sizes()[1:].sum()) > 1000
    assert gz.adapters_read1() == list(rz.adapters_read1()) and gz.adapters_read2() == list(rz.adapters_read2())
