"""The reference's own unit tests for the hot-path modules, run against
sequali_amd (same class names, calls and assertions as tests/test_qc_metrics.py,
test_adapter_counter.py, test_per_tile_quality.py, test_overrepresented_sequences.py,
test_dedup_estimator.py, test_insert_size_metrics.py of sequali v1.0.2; data files
replaced by the golden fixtures).  Needs a GPU."""
import ctypes
import itertools
import math
import string
import warnings

import numpy as np
import pytest

from tests.helpers import golden, split_fastq

pytestmark = pytest.mark.gpu


def records_of(name):
    buf, metas = split_fastq(golden(name)["fastq"].tobytes())
    out = []
    for m in metas:
        s = int(m["record_start"])
        seq = buf[s + int(m["sequence_offset"]):s + int(m["sequence_offset"]) + int(m["sequence_length"])].decode()
        qual = buf[s + int(m["qualities_offset"]):s + int(m["qualities_offset"]) + int(m["sequence_length"])].decode()
        out.append((seq, qual))
    return out


QC_CASES = [
    ("A" * 10 + "C" * 10 + "G" * 10 + "T" * 10 + "N" * 10, chr(43) * 25 + chr(63) * 25, 15),
    ("A" * 10 + "C" * 10 + "G" * 10 + "T" * 10 + "N" * 10, chr(43) * 25 + chr(63) * 25, 100),
    ("A" * 10 + "C" * 10 + "G" * 10 + "T" * 10 + "N" * 10, chr(43) * 25 + chr(63) * 25, 50),
    ("A" * 50, chr(34) * 25 + chr(38) * 25, 100),
] + [(s, q, 100) for s, q in records_of("ref_simple")] + \
    [(s, q, 100) for s, q in records_of("ref_100_nanopore")[:12]]


@pytest.mark.parametrize(["sequence", "qualities", "end_anchor_length"], QC_CASES)
def test_qc_metrics(sequence, qualities, end_anchor_length):
    """tests/test_qc_metrics.py:55-141"""
    from sequali_amd import A, C, G, N, T, NUMBER_OF_NUCS, NUMBER_OF_PHREDS, FastqRecordView, QCMetrics

    def base_to_index(base):
        return {"A": A, "C": C, "G": G, "T": T}.get(base.upper(), N)

    metrics = QCMetrics(end_anchor_length=end_anchor_length)
    assert metrics.end_anchor_length == end_anchor_length
    metrics.add_read(FastqRecordView("name", sequence, qualities))
    assert metrics.max_length == len(sequence)
    assert metrics.number_of_reads == 1
    gc_content = metrics.gc_content()
    assert sum(gc_content) == 1
    at_count = sum(sequence.upper().count(c) for c in "AT")
    gc_count = sum(sequence.upper().count(c) for c in "GC")
    assert gc_content[round((gc_count * 100) / (at_count + gc_count))] == 1
    phred_content = metrics.phred_scores()
    this_read_error = sum(10 ** -((ord(c) - 33) / 10) for c in qualities)
    this_read_phred = -10 * math.log10(this_read_error / len(sequence))
    assert phred_content[math.floor(this_read_phred)] == 1
    assert sum(phred_content) == 1
    phred_array = metrics.phred_count_table()
    assert len(phred_array) == len(sequence) * NUMBER_OF_PHREDS
    assert sum(phred_array) == len(sequence)
    for i, char in enumerate(qualities):
        assert phred_array[min((ord(char) - 33) // 4, 11) + NUMBER_OF_PHREDS * i] == 1
    base_array = metrics.base_count_table()
    assert len(base_array) == len(sequence) * NUMBER_OF_NUCS
    for i, nuc in enumerate(sequence):
        assert base_array[i * NUMBER_OF_NUCS + base_to_index(nuc)] == 1
    end_bases = metrics.end_anchored_base_count_table()
    end_phreds = metrics.end_anchored_phred_count_table()
    assert len(end_bases) == end_anchor_length * NUMBER_OF_NUCS
    end_sequence = sequence[max(len(sequence) - end_anchor_length, 0):]
    end_quals = qualities[max(len(sequence) - end_anchor_length, 0):]
    end_offset = max(end_anchor_length - len(sequence), 0)
    for i, base in enumerate(end_sequence):
        assert end_bases[(end_offset + i) * NUMBER_OF_NUCS + base_to_index(base)] == 1
    for i, phred in enumerate(end_quals):
        assert end_phreds[(end_offset + i) * NUMBER_OF_PHREDS + min((ord(phred) - 33) // 4, 11)] == 1


def test_long_sequence():
    """tests/test_qc_metrics.py:143-159"""
    from sequali_amd import FastqRecordView, QCMetrics
    metrics = QCMetrics()
    sequence = 4096 * "A" + 4096 * "C"
    qualities = 2048 * chr(33) + 2048 * chr(43) + 2048 * chr(53) + 2048 * chr(63)
    errors = 2048 * (10 ** -0) + 2048 * (10 ** -1) + 2048 * (10 ** -2) + 2048 * (10 ** -3)
    metrics.add_read(FastqRecordView("name", sequence, qualities))
    assert metrics.phred_scores()[math.floor(-10 * math.log10(errors / 8192))] == 1
    assert metrics.gc_content()[50] == 1


def test_average_long_quality():
    """tests/test_qc_metrics.py:162-173: a 20 Mbp read, 1000 x Q0 then Q50"""
    from sequali_amd import FastqRecordView, QCMetrics
    metrics = QCMetrics()
    n = 20_000_000
    metrics.add_read(FastqRecordView("name", n * "A", 1000 * chr(33) + (n - 1000) * chr(83)))
    error_rate = (1 + 19999 * 10 ** -5) / 20000
    assert metrics.phred_scores()[math.floor(-10 * math.log10(error_rate))] == 1


def test_qc_metrics_type_errors():
    from sequali_amd import QCMetrics
    m = QCMetrics()
    with pytest.raises(TypeError, match="FastqRecordView"):
        m.add_read(b"ACGT")
    with pytest.raises(TypeError, match="FastqRecordArrayView"):
        m.add_record_array([1, 2])


# ---- AdapterCounter: tests/test_adapter_counter.py ------------------------------------
def test_adapter_counter_basic_init():
    from sequali_amd import AdapterCounter
    adapters = ["GATTTAGAGACATA", "TATACCCGTACCACAGAT", "GCCCGGGAAATTAGGCACGATT",
                "GCAGAGAGATATAGAGATACACACAGAGAGAGAT", "GGGCACCACAGAGACCACACAGAGACA"]
    counter = AdapterCounter(adapters)
    assert counter.adapters == tuple(adapters)
    assert counter.max_length == 0
    assert counter.number_of_sequences == 0


def test_adapter_counter_init_errors():
    from sequali_amd import MAX_SEQUENCE_SIZE, AdapterCounter
    with pytest.raises(TypeError, match="not iterable"):
        AdapterCounter(1)
    with pytest.raises(ValueError, match="t least one"):
        AdapterCounter([])
    with pytest.raises(TypeError, match="b'GATTACA'"):
        AdapterCounter(["GATTACA", b"GATTACA"])
    with pytest.raises(ValueError, match="ASCII"):
        AdapterCounter(["GATTACA", "Gättaca"])
    with pytest.raises(ValueError, match=str(MAX_SEQUENCE_SIZE + 1)):
        AdapterCounter(["A" * 31, "A" * (MAX_SEQUENCE_SIZE + 1)])


def test_adapter_counter_matcher():
    from sequali_amd import AdapterCounter, FastqRecordView
    counter = AdapterCounter(["GATTACA", "GGGG", "TTTTT"])
    sequence = "AAGATTACAAAAAGATTACAGGGGAACGAGGGG"  # only the first match counts
    counter.add_read(FastqRecordView("bla", sequence, "H" * len(sequence)))
    counts = counter.get_counts()
    assert [c[0] for c in counts] == ["GATTACA", "GGGG", "TTTTT"]
    gattaca = counts[0][1].tolist()
    assert len(gattaca) == len(sequence) and sum(gattaca) == 1
    assert gattaca[sequence.find("GATTACA")] == 1
    gggg = counts[1][1].tolist()
    assert gggg[sequence.find("GGGG")] == 1 and sum(gggg) == 1
    assert sum(counts[2][1].tolist()) == 0


@pytest.mark.parametrize("adapters", [
    ["A" * 64, "C" * 64, "G" * 64, "T" * 64], ["A" * 64, "C" * 64, "G" * 64],
    ["A" * 64, "C" * 64], ["A" * 64, "C" * 64, "G" * 64, "T" * 64, "N" * 64]])
def test_adapter_counter_matcher_multiple_machine_words(adapters):
    from sequali_amd import AdapterCounter, FastqRecordView
    sequence = ("GATTACA" * 20).join(adapters)
    counter = AdapterCounter(adapters)
    counter.add_read(FastqRecordView("name", sequence, "H" * len(sequence)))
    for adapter, forward_counts, reverse_counts in counter.get_counts():
        index = sequence.find(adapter)
        assert forward_counts[index] == 1
        assert reverse_counts[len(sequence) - 1 - index] == 1
        assert sum(forward_counts) == 1 and sum(reverse_counts) == 1


# ---- PerTileQuality: tests/test_per_tile_quality.py -------------------------------------
def test_per_tile_quality():
    from sequali_amd import FastqRecordView, PerTileQuality
    ptq = PerTileQuality()
    ptq.add_read(FastqRecordView("SIM:1:FCX:1:15:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA", "AAAA", "ABCD"))
    assert ptq.number_of_reads == 1 and ptq.max_length == 4 and ptq.skipped_reason is None
    (tile, sum_list, count_list), = ptq.get_tile_counts()
    assert tile == 15
    assert sum_list == [10 ** (-32 / 10), 10 ** (-33 / 10), 10 ** (-34 / 10), 10 ** (-35 / 10)]
    assert count_list == [1, 1, 1, 1]


@pytest.mark.parametrize("tile_id", [0, 1, 9, 10, 99, 1234, 99239])
def test_tile_parse_correct(tile_id):
    from sequali_amd import FastqRecordView, PerTileQuality
    ptq = PerTileQuality()
    ptq.add_read(FastqRecordView(f"SIM:1:FCX:1:{tile_id}:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA", "AAAA", "ABCD"))
    assert ptq.get_tile_counts()[0][0] == tile_id


@pytest.mark.parametrize("header", [
    "SIMULATED_NAME", "SIM:1:FCX:1::6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA",
    "SIM:1:FCX:1:abc:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA",
    "SIM:1:FCX:1:0x1a3:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA", "SIM:1:FCX:1", "SIM:1:FCX:1:1045"])
def test_per_tile_quality_skip(header):
    from sequali_amd import FastqRecordView, PerTileQuality
    ptq = PerTileQuality()
    ptq.add_read(FastqRecordView(header, "AAAA", "ABCD"))
    assert ptq.number_of_reads == 0 and ptq.max_length == 0
    assert header in ptq.skipped_reason
    ptq.add_read(FastqRecordView("SIM:1:FCX:1:15:6329:1045 1:N:0:A", "AAAA", "ABCD"))  # stays off
    assert ptq.number_of_reads == 0


# ---- OverrepresentedSequences: tests/test_overrepresented_sequences.py ---------------------
def view_from_sequence(sequence):
    from sequali_amd import FastqRecordView
    return FastqRecordView("name", sequence, "A" * len(sequence))


def test_overrepresented_sequences_cap():
    """:33-60 with 4^6 reads and a cap of 1000 (the original uses 4^9 / 100 000)"""
    from sequali_amd import FastqRecordArrayView, OverrepresentedSequences
    cap, k, letters = 1000, 31, 6
    seqdup = OverrepresentedSequences(max_unique_fragments=cap, fragment_length=k, sample_every=1)
    reads = [view_from_sequence("".join(c) + (k - letters) * "A") for c in itertools.product("ACGT", repeat=letters)]
    for i in range(0, len(reads), 512):
        seqdup.add_record_array(FastqRecordArrayView(reads[i:i + 512]))
    assert seqdup.number_of_sequences == 4 ** letters
    counts = seqdup.sequence_counts()
    assert len(counts) == cap == seqdup.max_unique_fragments
    assert all(len(s) == k and c == 1 for s, c in counts.items())
    seqdup.add_read(view_from_sequence(k * "A"))
    assert seqdup.sequence_counts()[k * "A"] == 2


def test_overrepresented_sequences_overrepresented_sequences():
    """:83-119"""
    from sequali_amd import FastqRecordArrayView, OverrepresentedSequences
    k = 31
    seqs = OverrepresentedSequences(sample_every=1, fragment_length=k)
    reads = (["A" * k] * 100 + ["C" * k] * 200 + ["G" * k] * 2000 + ["T" * k] * 10 + ["C" * (k - 1) + "A"] +
             ["A" * (k - 1) + "C"] * (100_000 - 2311))
    views = [view_from_sequence(s) for s in set(reads)]
    cache = {v.sequence(): v for v in views}
    for i in range(0, len(reads), 20000):
        seqs.add_record_array(FastqRecordArrayView([cache[s] for s in reads[i:i + 20000]]))
    over = seqs.overrepresented_sequences(threshold_fraction=0.001)
    assert over[0][2] == "A" * (k - 1) + "C"
    assert over[1][2] == "C" * k and over[1][0] == 2200
    assert over[2][2] == "A" * k and over[2][1] == 110 / 100_000
    assert len(over) == 3
    assert seqs.overrepresented_sequences(threshold_fraction=0.00001)[-1][2] == "C" * (k - 1) + "A"
    assert seqs.overrepresented_sequences(threshold_fraction=0.00001, min_threshold=2)[-1][2] == "A" * k
    over = seqs.overrepresented_sequences(threshold_fraction=0.1, min_threshold=2, max_threshold=1000)
    assert len(over) == 2 and over[1][2] == "C" * k


@pytest.mark.parametrize("threshold", [-0.1, 1.1])
def test_overrepresented_faulty_threshold(threshold):
    from sequali_amd import OverrepresentedSequences
    with pytest.raises(ValueError, match="between"):
        OverrepresentedSequences().overrepresented_sequences(threshold_fraction=threshold)


def test_overrepresented_sequences_case_insensitive():
    from sequali_amd import OverrepresentedSequences
    k = 31
    seqs = OverrepresentedSequences(fragment_length=k, sample_every=1)
    seqs.add_read(view_from_sequence("aaTTaca" * 5))
    seqs.add_read(view_from_sequence("AAttACA" * 5))
    counts = seqs.sequence_counts()
    assert seqs.number_of_sequences == 2 and seqs.total_fragments == 4
    assert seqs.collected_unique_fragments == 2 and len(counts) == 2
    assert counts[("AATTACA" * 5)[:k]] == 2 and counts[("AATTACA" * 5)[-k:]] == 2


@pytest.mark.parametrize("divisor", [1, 2, 3, 7, 8, 20])
def test_overrepresented_sequences_sampling_rate(divisor):
    from sequali_amd import FastqRecordArrayView, OverrepresentedSequences
    seqs = OverrepresentedSequences(sample_every=divisor)
    read = view_from_sequence("AAAA")
    for chunk in (3000, 1, 6999):      # sampling phase carries over between arrays
        seqs.add_record_array(FastqRecordArrayView([read] * chunk))
    assert seqs.number_of_sequences == 10_000
    assert seqs.sampled_sequences == (10_000 + divisor - 1) // divisor


@pytest.mark.parametrize(["sequence", "result"], [
    ("GATTACAGATTACA", {"ATC": 1, "GTA": 1, "AGA": 1, "ACA": 1, "AAT": 1}),
    ("GATTACAAA", {"ATC": 1, "GTA": 1, "AAA": 1}), ("GA", {}), ("GATT", {"ATC": 1, "AAT": 1}),
    ("GATTACGATTAC", {"ATC": 1, "GTA": 1}), ("ACT", {"ACT": 1})])
def test_overrepresented_sequences_all_fragments(sequence, result):
    from sequali_amd import OverrepresentedSequences
    seqs = OverrepresentedSequences(fragment_length=3, sample_every=1)
    seqs.add_read(view_from_sequence(sequence))
    assert seqs.sequence_counts() == result


def test_non_iupac_warning_and_n_does_not_warn():
    from sequali_amd import OverrepresentedSequences
    seqs = OverrepresentedSequences(fragment_length=3, sample_every=1)
    with pytest.warns(UserWarning, match="KKK"):
        seqs.add_read(view_from_sequence("KKK"))
    seqs = OverrepresentedSequences(fragment_length=3, sample_every=1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        seqs.add_read(view_from_sequence("ACGTN"))
    assert seqs.sampled_sequences == 1


@pytest.mark.parametrize(["bases_from_start", "bases_from_end", "result"], [
    (0, 0, ()), (1, 1, ("AAC", "CAA")), (3, 3, ("AAC", "CAA")), (4, 4, ("AAC", "CAA", "CCG", "GCC")),
    (1, 0, ("AAC",)), (0, 1, ("CAA",)), (100, 100, ("AAA", "AAC", "CAA", "CCG", "GCC")),
    (-1, -1, ("AAA", "AAC", "CAA", "CCG", "GCC"))])
def test_overrepresented_sequences_sample_from_begin_and_end(bases_from_start, bases_from_end, result):
    from sequali_amd import OverrepresentedSequences
    seqs = OverrepresentedSequences(fragment_length=3, sample_every=1, bases_from_start=bases_from_start,
                                    bases_from_end=bases_from_end)
    seqs.add_read(view_from_sequence("AACCGGTTTTGGCCAA"))
    assert tuple(sorted(x[2] for x in seqs.overrepresented_sequences(min_threshold=1))) == result


# ---- DedupEstimator: tests/test_dedup_estimator.py -------------------------------------------
def test_dedup_estimator():
    from sequali_amd import DedupEstimator
    dedup_est = DedupEstimator(160)
    assert dedup_est._hash_table_size == 1 << 8
    for s in ("test", "test2", "test3", "test4"):
        dedup_est.add_sequence(s)
    for _ in range(100):
        dedup_est.add_sequence("test5")
    dupcounts = sorted(dedup_est.duplication_counts())
    assert len(dupcounts) == dedup_est.tracked_sequences
    assert dupcounts[-1] == 100 and dupcounts[0] == 1


def test_dedup_estimator_switches_modulo():
    from sequali_amd import DedupEstimator
    from sequali_amd._qc import _array_of_sequences
    dedup_est = DedupEstimator(179)
    assert dedup_est._modulo_bits == 0
    seqs = ["".join(x) for _, x in zip(range(10000), itertools.product(*([string.ascii_letters] * 10)))]
    for i in range(0, 10000, 1000):
        dedup_est.add_record_array(_array_of_sequences(seqs[i:i + 1000]))
    assert dedup_est._modulo_bits == 6


@pytest.mark.parametrize("parameter,value", [("max_stored_fingerprints", 7), ("front_sequence_length", -1),
                                             ("back_sequence_length", -1), ("front_sequence_offset", -1),
                                             ("back_sequence_offset", -1)])
def test_dedup_estimator_invalid_settings(parameter, value):
    from sequali_amd import DedupEstimator
    with pytest.raises(ValueError) as e:
        DedupEstimator(**{parameter: value})
    assert e.match(parameter) and e.match(str(value))


@pytest.mark.parametrize("fl,fo,bl,bo", [(1, 0, 1, 0), (8, 64, 8, 64), (100000, 80000, 10000, 80000),
                                         (1, 0, 0, 0), (0, 0, 1, 0)])
def test_dedup_estimator_valid_settings(fl, fo, bl, bo):
    from sequali_amd import DedupEstimator
    d = DedupEstimator(front_sequence_length=fl, front_sequence_offset=fo, back_sequence_length=bl,
                       back_sequence_offset=bo)
    d.add_sequence("test")
    d.add_sequence("test2")
    assert sum(d.duplication_counts()) == 2


SIX = ["123456AC TA123451", "234561AC AA234561", "345612AC TA345611",
       "456123AG AA456121", "561234AG TA561231", "612345AG AA612341"]


@pytest.mark.parametrize("fl,fo,bl,bo,result", [(8, 0, 8, 0, {1}), (0, 0, 6, 0, {1}), (1, 6, 1, 6, {6}),
                                                (2, 6, 1, 6, {3}), (2, 6, 2, 6, {2, 1}), (1, 0, 0, 0, {1}),
                                                (0, 0, 1, 0, {6})])
def test_dedup_estimator_offsets_and_lengths(fl, fo, bl, bo, result):
    from sequali_amd import DedupEstimator
    d = DedupEstimator(front_sequence_offset=fo, front_sequence_length=fl, back_sequence_length=bl,
                       back_sequence_offset=bo, max_stored_fingerprints=100)
    for s in SIX:
        d.add_sequence(s)
    assert set(d.duplication_counts()) == result


@pytest.mark.parametrize("fl,fo,bl,bo,result", [(8, 0, 8, 0, {1}), (0, 0, 6, 0, {1}), (1, 6, 1, 1, {6}),
                                                (2, 6, 1, 1, {3}), (2, 6, 2, 0, {2, 1}), (1, 0, 0, 0, {1}),
                                                (0, 0, 1, 7, {6})])
def test_dedup_estimator_offsets_and_lengths_paired(fl, fo, bl, bo, result):
    from sequali_amd import DedupEstimator
    d = DedupEstimator(front_sequence_offset=fo, front_sequence_length=fl, back_sequence_length=bl,
                       back_sequence_offset=bo, max_stored_fingerprints=100)
    for s in SIX:
        a, b = s.split()
        d.add_sequence_pair(a, b)
    assert set(d.duplication_counts()) == result


# ---- InsertSizeMetrics: tests/test_insert_size_metrics.py ---------------------------------------
R1, R2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


@pytest.mark.parametrize(["sequence1", "sequence2", "insert_size"], [
    ("ATATATATATATATAT", "ATATATATATATATAT", 16),
    ("ATATATATATATATATNNNNNNNNNN", "ATATATATATATATATNNNNNNNNNN", 16),
    ("NNNNNNNNNNATATATATATATATAT", "ATATATATATATATATNNNNNNNNNN", 26),
    ("ACGTTGCAGCTATCGA" + R1, "TCGATAGCTGCAACGT" + R2, 16),
    ("GTACACGTTGCAGCTATCGA" + R1, "TCGATAGCTGCAACGTGTAC" + R2, 20),
    ("GTACACGTTGCAGCTATCGA" + R1, "tcgatagctgcaacgtgtac" + R2, 20),
    ("GTACACGTTGCAGCTATCGA" + R1, "tcGatagCTgcaAcgtGtac" + R2, 20)])
def test_insert_size_metrics(sequence1, sequence2, insert_size):
    from sequali_amd import INSERT_SIZE_MAX_ADAPTER_STORE_SIZE, InsertSizeMetrics
    ism = InsertSizeMetrics()
    ism.add_sequence_pair(sequence1, sequence2)
    assert ism.insert_sizes()[insert_size] == 1
    for seq, table, n in ((sequence1, ism.adapters_read1(), ism.number_of_adapters_read1),
                          (sequence2, ism.adapters_read2(), ism.number_of_adapters_read2)):
        adapter = seq[insert_size:][:INSERT_SIZE_MAX_ADAPTER_STORE_SIZE]
        if adapter:
            assert dict(table).get(adapter) == 1 and n == 1
        else:
            assert n == 0


# ---- the raw C ABI with host buffers (what a C caller binds) ------------
def test_c_abi_host_buffer_entry_points():
    from oracle import oracle
    from sequali_amd import _lib, synth
    lib, ctx = _lib.lib(), _lib.context()
    buf, metas = synth.host_records(synth.ILLUMINA, 3, 777)
    raw = np.frombuffer(buf, dtype=np.uint8)
    ref = oracle.QCMetrics()
    ref_metas = metas.copy()
    ref.add(buf, ref_metas)
    h = lib.sq_qcmetrics_new(ctx, 100)
    assert lib.sq_qcmetrics_add(h, raw.ctypes.data, len(raw), metas.ctypes.data, len(metas)) == 0
    # accumulated_error_rate came back into the caller's metas (_qcmodule.c:2126)
    np.testing.assert_array_equal(metas["accumulated_error_rate"].view(np.uint64),
                                  ref_metas["accumulated_error_rate"].view(np.uint64))
    n = lib.sq_qcmetrics_base_count_table(h, None, 0)
    out = np.zeros(n, np.uint64)
    assert lib.sq_qcmetrics_base_count_table(h, out.ctypes.data, n) == n
    np.testing.assert_array_equal(out, ref.base_count_table())
    # invalid phred byte: negative return code and the reference's message
    bad_buf, bad_metas = oracle.make_batch(["x"], ["ACGT"], ["II~\x7f"])
    bad = np.frombuffer(bad_buf, dtype=np.uint8)
    rc = lib.sq_qcmetrics_add(h, bad.ctypes.data, len(bad), bad_metas.ctypes.data, 1)
    assert rc == -2 and _lib.last_error() == "Not a valid phred character: \x7f"
    lib.sq_qcmetrics_free(h)

    probes = [p.encode() for p in synth.ILLUMINA_PROBES]
    ptrs = (ctypes.c_char_p * len(probes))(*probes)
    lens = (ctypes.c_size_t * len(probes))(*[len(p) for p in probes])
    a = lib.sq_adaptercounter_new(ctx, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(lens, ctypes.c_void_p), len(probes))
    assert lib.sq_adaptercounter_add(a, raw.ctypes.data, len(raw), metas.ctypes.data, len(metas)) == 0
    ra = oracle.AdapterCounter(list(synth.ILLUMINA_PROBES))
    ra.add(buf, metas)
    f, r = np.zeros(150, np.uint64), np.zeros(150, np.uint64)
    for i, (_, fr, rr) in enumerate(ra.get_counts()):
        assert lib.sq_adaptercounter_get_counts(a, i, f.ctypes.data, r.ctypes.data, 150) == 150
        np.testing.assert_array_equal(f, fr)
        np.testing.assert_array_equal(r, rr)
    lib.sq_adaptercounter_free(a)


# ---- tests/test_nano_stats.py of the reference ----------------------------------------
def test_nano_stats_from_header():
    import datetime
    from sequali_amd import FastqRecordView, NanoStats
    view = FastqRecordView("cb1dab45-aa4c-43fc-a91e-ad0ecc92f5c9 "
                           "runid=c989c681b782549923cb0a02c95f6ec9d2534335 "
                           "read=10 ch=444 start_time=2021-09-30T11:34:08Z "
                           "flow_cell_id=PAI09842 protocol_group_id=SS_210930_10xCDNA sample_id=SS_A1",
                           "ACGT", "AAAA")
    timestamp = datetime.datetime(2021, 9, 30, 11, 34, 8, tzinfo=datetime.timezone.utc).timestamp()
    nanostats = NanoStats()
    nanostats.add_read(view)
    assert nanostats.minimum_time == timestamp and nanostats.maximum_time == timestamp
    infos = list(nanostats.nano_info_iterator())
    assert len(infos) == 1
    info = infos[0]
    assert info.start_time == timestamp and info.channel_id == 444 and info.length == 4
    assert info.cumulative_error_rate == 4 * 10 ** (-(ord("A") - 33) / 10)
    assert info.duration == 0.0


def test_nano_stats_from_tags():
    import datetime
    import struct
    from sequali_amd import FastqRecordView, NanoStats
    timestamp_string, readgroup_string = b"2021-09-30T11:34:08Z\x00", b"SS_A1\x00"
    parent_id = b"8D8AC610-566D-4EF0-9C22-186B2A5ED793\x00"
    tags = struct.pack(f"<3sB3sH3s{len(timestamp_string)}s3s{len(readgroup_string)}s3sf3s{len(parent_id)}s",
                       b"rnC", 10, b"chS", 444, b"stZ", timestamp_string, b"RGZ", readgroup_string,
                       b"duf", 2.5, b"piZ", parent_id)
    view = FastqRecordView("cb1dab45-aa4c-43fc-a91e-ad0ecc92f5c9", "ACGT", "AAAA", tags)
    timestamp = datetime.datetime(2021, 9, 30, 11, 34, 8, tzinfo=datetime.timezone.utc).timestamp()
    nanostats = NanoStats()
    nanostats.add_read(view)
    assert nanostats.minimum_time == timestamp and nanostats.maximum_time == timestamp
    info, = list(nanostats.nano_info_iterator())
    assert info.start_time == timestamp and info.channel_id == 444 and info.length == 4
    assert info.cumulative_error_rate == 4 * 10 ** (-(ord("A") - 33) / 10)
    assert info.duration == 2.5
    assert info.parent_id_hash == int(parent_id[:8] + parent_id[-9:-1], 16)
