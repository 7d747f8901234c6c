#!/usr/bin/env python3
"""Capture golden vectors from the compiled reference (oracle/_ref/_qc.abi3.so).

Run in the build container only (needs oracle/_ref, built by `make -C oracle`
from /root/reference, and the reference's tests/data for the file cases):

    python tests/golden/make_golden.py

Output: tests/golden/*.npz -- for every case the FASTQ text that was fed
(input, as data) and every getter output of the reference's modules
(expected).  The reference itself never travels; these files do.
"""
from __future__ import annotations

import ctypes
import gzip
import io
import itertools
import json
import os
import string
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.path.insert(0, ROOT)
import _qc  # noqa: E402  (the reference build)

REFDATA = "/root/reference/tests/data"

ILLUMINA_PROBES = ["AGATCGGAAGAG", "TGGAATTCTCGG", "GATCGTCGGACT", "CTGTCTCTTATA",
                   "GGGGGGGGGGGG", "AAAAAAAAAAAA"]  # adapter_list.tsv:8-15
NANOPORE_PROBES = ["TTACGTATTGCT", "GCAATACGTAAC", "CTTGCGGGCGGC", "GGTAGTAGGTTC",
                   "GAGGCGAGCGGT", "CAAGATACGCAC", "GTGACTTGCCTG", "ATCGCCTACCGT",
                   "TCTATCTTCTTT", "TCTTCAGAGGAG", "GATATTGCTGGG", "TGATATTGCTTT",
                   "GTACGTATTGCT", "ACGTAACTGAAC"]  # adapter_list.tsv:35-57


def read_file(name: str) -> bytes:
    path = os.path.join(REFDATA, name)
    opener = gzip.open if name.endswith(".gz") else open
    with opener(path, "rb") as f:
        return f.read()


def fastq_text(records) -> bytes:
    return b"".join(b"@%s\n%s\n+\n%s\n" % (n.encode(), s.encode(), q.encode())
                    for n, s, q in records)


def arrays_of(text: bytes, buffersize: int = 128 * 1024):
    """The reference's own parser, its own chunking."""
    return list(_qc.FastqParser(io.BytesIO(text), buffersize))


def error_rates_of(arr) -> np.ndarray:
    """accumulated_error_rate of every FastqMeta in a reference array object
    (object layout: SURVEY 8b interop note; ob_size@16, records@32, 40 B each)."""
    n = len(arr)
    raw = (ctypes.c_char * (n * 40)).from_address(id(arr) + 32)
    return np.frombuffer(bytes(raw), dtype=np.dtype([("pad", "V32"), ("err", "<f8")]))["err"].copy()


def qc_outputs(arrays, end_anchor=100, prefix="qc_"):
    m = _qc.QCMetrics(end_anchor)
    errs = []
    for a in arrays:
        m.add_record_array(a)
        errs.append(error_rates_of(a))
    return {
        prefix + "base": np.array(m.base_count_table(), np.uint64),
        prefix + "phred": np.array(m.phred_count_table(), np.uint64),
        prefix + "ea_base": np.array(m.end_anchored_base_count_table(), np.uint64),
        prefix + "ea_phred": np.array(m.end_anchored_phred_count_table(), np.uint64),
        prefix + "gc": np.array(m.gc_content(), np.uint64),
        prefix + "phred_scores": np.array(m.phred_scores(), np.uint64),
        prefix + "number_of_reads": np.uint64(m.number_of_reads),
        prefix + "max_length": np.uint64(m.max_length),
        prefix + "end_anchor": np.uint64(end_anchor),
        prefix + "error_rates": np.concatenate(errs) if errs else np.zeros(0),
    }


def adapter_outputs(arrays, probes, prefix="ad_"):
    c = _qc.AdapterCounter(probes)
    for a in arrays:
        c.add_record_array(a)
    counts = c.get_counts()
    return {
        prefix + "probes": np.array(probes),
        prefix + "fwd": np.array([np.array(f, np.uint64) for _, f, _ in counts], np.uint64).reshape(len(probes), -1),
        prefix + "rev": np.array([np.array(r, np.uint64) for _, _, r in counts], np.uint64).reshape(len(probes), -1),
        prefix + "max_length": np.uint64(c.max_length),
        prefix + "number_of_sequences": np.uint64(c.number_of_sequences),
    }


def pertile_outputs(arrays, prefix="pt_"):
    p = _qc.PerTileQuality()
    for a in arrays:
        p.add_record_array(a)
    tc = p.get_tile_counts()
    ml = p.max_length
    return {
        prefix + "tiles": np.array([t for t, _, _ in tc], np.int64),
        prefix + "errors": np.array([e for _, e, _ in tc], np.float64).reshape(len(tc), ml),
        prefix + "counts": np.array([c for _, _, c in tc], np.uint64).reshape(len(tc), ml),
        prefix + "max_length": np.uint64(ml),
        prefix + "number_of_reads": np.uint64(p.number_of_reads),
        prefix + "skipped_reason": np.array(p.skipped_reason or ""),
    }


def overrep_outputs(arrays, prefix="ov_", **kw):
    import warnings
    o = _qc.OverrepresentedSequences(**kw)
    nwarn = 0
    for a in arrays:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            o.add_record_array(a)
            nwarn += len(w)
    sc = o.sequence_counts()
    keys = sorted(sc)
    ovr = o.overrepresented_sequences()
    return {
        prefix + "kwargs": np.array(json.dumps(kw)),
        prefix + "seqs": np.array(keys),
        prefix + "counts": np.array([sc[k] for k in keys], np.uint64),
        prefix + "number_of_sequences": np.uint64(o.number_of_sequences),
        prefix + "sampled_sequences": np.uint64(o.sampled_sequences),
        prefix + "total_fragments": np.uint64(o.total_fragments),
        prefix + "collected_unique_fragments": np.uint64(o.collected_unique_fragments),
        prefix + "ovr_counts": np.array([c for c, _, _ in ovr], np.uint64),
        prefix + "ovr_fracs": np.array([f for _, f, _ in ovr], np.float64),
        prefix + "ovr_seqs": np.array([s for _, _, s in ovr]),
    }


def dedup_outputs(arrays, arrays2=None, prefix="dd_", **kw):
    d = _qc.DedupEstimator(**kw)
    if arrays2 is None:
        for a in arrays:
            d.add_record_array(a)
    else:
        for a, b in zip(arrays, arrays2):
            d.add_record_array_pair(a, b)
    return {
        prefix + "kwargs": np.array(json.dumps(kw)),
        prefix + "counts_slot_order": np.array(d.duplication_counts(), np.uint64),
        prefix + "modulo_bits": np.uint64(d._modulo_bits),
        prefix + "tracked_sequences": np.uint64(d.tracked_sequences),
        prefix + "hash_table_size": np.uint64(d._hash_table_size),
    }


def insert_outputs(arrays1, arrays2, prefix="is_", **kw):
    z = _qc.InsertSizeMetrics(**kw)
    for a, b in zip(arrays1, arrays2):
        z.add_record_array_pair(a, b)
    a1, a2 = z.adapters_read1(), z.adapters_read2()
    return {
        prefix + "kwargs": np.array(json.dumps(kw)),
        prefix + "insert_sizes": np.array(z.insert_sizes(), np.uint64),
        prefix + "total_reads": np.uint64(z.total_reads),
        prefix + "n_adapters_read1": np.uint64(z.number_of_adapters_read1),
        prefix + "n_adapters_read2": np.uint64(z.number_of_adapters_read2),
        prefix + "ad1_seqs": np.array([s for s, _ in a1]), prefix + "ad1_counts": np.array([c for _, c in a1], np.uint64),
        prefix + "ad2_seqs": np.array([s for s, _ in a2]), prefix + "ad2_counts": np.array([c for _, c in a2], np.uint64),
    }


def paired_arrays(text1: bytes, text2: bytes):
    """Lock-step reading like __main__.py:279-285."""
    p1 = _qc.FastqParser(io.BytesIO(text1))
    p2 = _qc.FastqParser(io.BytesIO(text2))
    a1, a2 = [], []
    for arr in p1:
        a1.append(arr)
        a2.append(p2.read(len(arr)))
    return a1, a2


def save(name: str, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in list(arrays.items())[:4]}, "...")


def text_entry(key: str, text: bytes, gen=None) -> dict:
    """The input text of a fixture: the bytes, or -- for slices of the build's own counter-based
    generator -- (kind, first, n, seed) and the SHA-256 of the bytes the reference was run on
    (tests/helpers.py golden_text regenerates and checks them)."""
    if gen is None:
        return {key: np.frombuffer(text, np.uint8)}
    import hashlib
    return {key + "_gen": np.array(gen, dtype=np.int64),
            key + "_sha256": np.frombuffer(hashlib.sha256(text).digest(), np.uint8)}


def single_end_case(name: str, text: bytes, probes, gen=None, **extra):
    arrays = arrays_of(text)
    out = text_entry("fastq", text, gen)
    out.update(qc_outputs(arrays))
    out.update(adapter_outputs(arrays, probes))
    out.update(pertile_outputs(arrays))
    out.update(overrep_outputs(arrays))
    out.update(overrep_outputs(arrays, prefix="ov1_", sample_every=1))
    out.update(overrep_outputs(arrays, prefix="ovcap_", sample_every=1, max_unique_fragments=50,
                               fragment_length=11, bases_from_start=-1, bases_from_end=-1))
    out.update(dedup_outputs(arrays))
    out.update(dedup_outputs(arrays, prefix="dd0_", front_sequence_offset=64, back_sequence_offset=0))
    out.update(dedup_outputs(arrays, prefix="ddcap_", max_stored_fingerprints=100,
                             front_sequence_length=5, back_sequence_length=4,
                             front_sequence_offset=3, back_sequence_offset=2))
    out.update(extra)
    save(name, **out)


def main():
    # (5) SCORE_TO_ERROR_RATE bit patterns, read back through PerTileQuality
    bits = []
    for q in range(94):
        p = _qc.PerTileQuality()
        p.add_read(_qc.FastqRecordView("a:b:c:d:7:x", "A", chr(q + 33)))
        bits.append(int(np.float64(p.get_tile_counts()[0][1][0]).view(np.uint64)))
    with open(os.path.join(HERE, "error_table.json"), "wt") as f:
        json.dump(["0x%016x" % b for b in bits], f, indent=0)

    # (1) the reference's own test data
    single_end_case("ref_simple", read_file("simple.fastq"), ILLUMINA_PROBES)
    single_end_case("ref_100_illumina_adapters", read_file("100_illumina_adapters.fastq"), ILLUMINA_PROBES)
    single_end_case("ref_100_nanopore", read_file("100_nanopore_reads.fastq.gz"), NANOPORE_PROBES)
    for small in ("empty.fastq", "empty_read.fastq", "single_nuc.fastq",
                  "single_illumina_metadata.fastq"):
        single_end_case("ref_" + small.replace(".fastq", ""), read_file(small), ILLUMINA_PROBES)

    t1 = read_file("LTB-A-BC001_S1_L003_R1_001.fastq.gz")
    t2 = read_file("LTB-A-BC001_S1_L003_R2_001.fastq.gz")
    a1, a2 = paired_arrays(t1, t2)
    out = {"fastq1": np.frombuffer(t1, np.uint8), "fastq2": np.frombuffer(t2, np.uint8)}
    out.update(qc_outputs(a1, prefix="qc1_"))
    out.update(qc_outputs(a2, prefix="qc2_"))
    out.update(pertile_outputs(a1, prefix="pt1_"))
    out.update(pertile_outputs(a2, prefix="pt2_"))
    out.update(overrep_outputs(a1, prefix="ov1_"))
    out.update(overrep_outputs(a2, prefix="ov2_"))
    out.update(adapter_outputs(a1, ILLUMINA_PROBES, prefix="ad1_"))
    out.update(dedup_outputs(a1, a2, prefix="dd_", front_sequence_offset=0, back_sequence_offset=0))
    out.update(dedup_outputs(a1, a2, prefix="ddcap_", max_stored_fingerprints=100,
                             front_sequence_offset=0, back_sequence_offset=0))
    out.update(insert_outputs(a1, a2))
    out.update(insert_outputs(a1, a2, prefix="iscap_", max_adapters=16))
    out["is_mate"] = np.array([x.is_mate(y) for x, y in zip(a1, a2)])
    save("ref_LTB_paired", **out)

    # (2) inline cases of the reference's tests, re-expressed as data
    cases = [
        ("A" * 10 + "C" * 10 + "G" * 10 + "T" * 10 + "N" * 10, chr(43) * 25 + chr(63) * 25),
        ("A" * 50, chr(34) * 25 + chr(38) * 25),
        (4096 * "A" + 4096 * "C", 2048 * chr(33) + 2048 * chr(43) + 2048 * chr(53) + 2048 * chr(63)),
        ("acgtnACGTNKkRr-*", "!\"#$%&'()*+,-./~"),
        ("", ""),
    ]
    for ea in (15, 50, 100, 0):
        for i, (s, q) in enumerate(cases):
            text = fastq_text([("name", s, q)])
            save(f"inline_qc_{i}_ea{ea}", fastq=np.frombuffer(text, np.uint8),
                 **qc_outputs(arrays_of(text), end_anchor=ea))

    # test_average_long_quality (tests/test_qc_metrics.py:162-173), scaled to 2 Mbp
    # to keep the fixture small; the 20 Mbp original runs in the live comparison
    n = 2_000_000
    text = fastq_text([("name", n * "A", 1000 * chr(33) + (n - 1000) * chr(83))])
    o = qc_outputs(arrays_of(text, 8 * 1024 * 1024))
    save("inline_qc_long_quality", n=np.uint64(n), qc_gc=o["qc_gc"],
         qc_phred_scores=o["qc_phred_scores"], qc_error_rates=o["qc_error_rates"])

    # (4) H1 grid: uniform-quality reads, 94 qualities x 7 lengths
    lengths = [1, 2, 5, 36, 100, 150, 151]
    grid = np.zeros((94, len(lengths)), np.int64)
    errs = np.zeros((94, len(lengths)), np.float64)
    for q in range(94):
        for j, L in enumerate(lengths):
            text = fastq_text([("n", "A" * L, chr(q + 33) * L)])
            o = qc_outputs(arrays_of(text))
            grid[q, j] = int(np.nonzero(o["qc_phred_scores"])[0][0])
            errs[q, j] = o["qc_error_rates"][0]
    save("h1_uniform_quality_grid", lengths=np.array(lengths), bins=grid, error_rates=errs)

    # adapter counter: multi-word cases (tests/test_adapter_counter.py:98-166)
    for i, probes in enumerate([
            ["A" * 64, "C" * 64, "G" * 64, "T" * 64],
            ["A" * 64, "C" * 64, "G" * 64],
            ["A" * 64, "C" * 64],
            ["A" * 64, "C" * 64, "G" * 64, "T" * 64, "N" * 64]]):
        seq = ("GATTACA" * 20).join(probes)
        text = fastq_text([("name", seq, "H" * len(seq))])
        save(f"inline_adapter_words_{i}", fastq=np.frombuffer(text, np.uint8),
             **adapter_outputs(arrays_of(text), probes))
    mixed = ["TATAAATATAAATATAAA", "GATTACAGATTACAGATTACA", "AAAAAAAAAAAA", "TTTTTTTTTTTT",
             "CACGTCAGTTACCGGATAGA", "GGTCAAGGGGTAAATGATAT", "AGGTAGATTTATTTTATTTAT", "GGGTGGGAGGCC"]
    seqs = ["NNNNN".join(mixed[i] for i in range(8)),
            "NNNNNNN".join(mixed[i] for i in (1, 2, 4, 5, 6, 7)),
            "NNN".join(mixed[i] for i in (7, 2, 6, 3, 4, 7)),
            "AAGATTACAAAAAGATTACAGGGGAACGAGGGG", "nnnnKKKKggtcaaggggtaaatgatat"]
    text = fastq_text([("name", s, "H" * len(s)) for s in seqs])
    save("inline_adapter_mixed", fastq=np.frombuffer(text, np.uint8),
         **adapter_outputs(arrays_of(text), mixed + ["GATTACA", "GGGG", "TTTTT", "NNNN", "KK"]))

    # per tile: malformed headers (tests/test_per_tile_quality.py:66-85) after good ones
    good = [(f"SIM:1:FCX:1:{t}:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA", "ACGT" * (1 + t % 3), "ABCD" * (1 + t % 3))
            for t in list(range(100)) + [1234, 99239]]
    bad_headers = ["SIMULATED_NAME", "SIM:1:FCX:1::6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA",
                   "SIM:1:FCX:1:abc:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA",
                   "SIM:1:FCX:1:0x1a3:6329:1045:GATTACT+GTCTTAAC 1:N:0:ATCCGA",
                   "SIM:1:FCX:1", "SIM:1:FCX:1:1045", "a:b:c:d:1234567890123456789:e"]
    save("inline_pertile_good", fastq=np.frombuffer(fastq_text(good), np.uint8),
         **pertile_outputs(arrays_of(fastq_text(good))))
    for i, h in enumerate(bad_headers):
        recs = good[:7] + [(h, "AAAA", "ABCD")] + good[7:20]
        text = fastq_text(recs)
        save(f"inline_pertile_bad_{i}", fastq=np.frombuffer(text, np.uint8),
             **pertile_outputs(arrays_of(text)))

    # overrepresented: the small parametrised cases (tests/test_overrepresented_sequences.py:147-208)
    for i, s in enumerate(["GATTACAGATTACA", "GATTACAAA", "GA", "GATT", "GATTACGATTAC", "ACT", "KKK", "ACGTN"]):
        text = fastq_text([("name", s, "A" * len(s))])
        save(f"inline_overrep_k3_{i}", fastq=np.frombuffer(text, np.uint8),
             **overrep_outputs(arrays_of(text), fragment_length=3, sample_every=1))
    text = fastq_text([("name", "AACCGGTTTTGGCCAA", "A" * 16)])
    for i, (bs, be) in enumerate([(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (1, 0), (0, 1), (100, 100), (-1, -1)]):
        save(f"inline_overrep_ends_{i}", fastq=np.frombuffer(text, np.uint8),
             **overrep_outputs(arrays_of(text), fragment_length=3, sample_every=1,
                               bases_from_start=bs, bases_from_end=be))
    # cap crossing (tests/test_overrepresented_sequences.py:33-60, scaled: 4^6 reads, cap 1000)
    recs = [("n", "".join(c) + 25 * "A", "H" * 31) for c in itertools.product("ACGT", repeat=6)]
    recs.append(("n", 31 * "A", 31 * "A"))
    text = fastq_text(recs)
    save("inline_overrep_cap", fastq=np.frombuffer(text, np.uint8),
         **overrep_outputs(arrays_of(text), max_unique_fragments=1000, fragment_length=31, sample_every=1))

    # dedup: modulo switching (tests/test_dedup_estimator.py:41-53)
    ten = [string.ascii_letters] * 10
    seqs = ["".join(x) for _, x in zip(range(10000), itertools.product(*ten))]
    text = fastq_text([("n", s, "A" * len(s)) for s in seqs])
    for cap in (100, 137, 179, 500):
        save(f"inline_dedup_cap{cap}", fastq=np.frombuffer(text, np.uint8),
             **dedup_outputs(arrays_of(text), max_stored_fingerprints=cap))
    six = ["123456AC TA123451", "234561AC AA234561", "345612AC TA345611",
           "456123AG AA456121", "561234AG TA561231", "612345AG AA612341"]
    text = fastq_text([("n", s, "A" * len(s)) for s in six])
    text1 = fastq_text([("n", s.split()[0], "A" * 8) for s in six])
    text2 = fastq_text([("n", s.split()[1], "A" * 8) for s in six])
    a1, a2 = paired_arrays(text1, text2)
    for i, (fl, fo, bl, bo) in enumerate([(8, 0, 8, 0), (0, 0, 6, 0), (1, 6, 1, 6), (2, 6, 1, 6),
                                          (2, 6, 2, 6), (1, 0, 0, 0), (0, 0, 1, 0), (1, 6, 1, 1),
                                          (2, 6, 2, 0), (0, 0, 1, 7)]):
        kw = dict(front_sequence_length=fl, front_sequence_offset=fo, back_sequence_length=bl,
                  back_sequence_offset=bo, max_stored_fingerprints=100)
        save(f"inline_dedup_geom_{i}", fastq=np.frombuffer(text, np.uint8),
             fastq1=np.frombuffer(text1, np.uint8), fastq2=np.frombuffer(text2, np.uint8),
             **dedup_outputs(arrays_of(text), **kw),
             **dedup_outputs(a1, a2, prefix="ddp_", **kw))

    # insert size (tests/test_insert_size_metrics.py:24-80)
    R1, R2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
    pairs = [("ATATATATATATATAT", "ATATATATATATATAT"),
             ("ATATATATATATATATNNNNNNNNNN", "ATATATATATATATATNNNNNNNNNN"),
             ("NNNNNNNNNNATATATATATATATAT", "ATATATATATATATATNNNNNNNNNN"),
             ("ACGTTGCAGCTATCGA" + R1, "TCGATAGCTGCAACGT" + R2),
             ("GTACACGTTGCAGCTATCGA" + R1, "TCGATAGCTGCAACGTGTAC" + R2),
             ("GTACACGTTGCAGCTATCGA" + R1, "tcgatagctgcaacgtgtac" + R2),
             ("GTACACGTTGCAGCTATCGA" + R1, "tcGatagCTgcaAcgtGtac" + R2),
             ("gtacacgttgcagctatcga" + R1, "TCGATAGCTGCAACGTGTAC" + R2),   # lowercase R1 (Q8)
             ("GTACACGTTGCAGCTATCGT" + R1, "TCGATAGCTGCAACGTGTAC" + R2),   # one mismatch
             ("GTACACGTTGCAGCTATGGT" + R1, "TCGATAGCTGCAACGTGTAC" + R2),   # two mismatches
             ("ACGT", "ACGT"), ("A" * 15, "T" * 40), ("", "")]
    text1 = fastq_text([("n", a, "A" * len(a)) for a, _ in pairs])
    text2 = fastq_text([("n", b, "A" * len(b)) for _, b in pairs])
    for i in range(len(pairs)):
        one1 = fastq_text([("n", pairs[i][0], "A" * len(pairs[i][0]))])
        one2 = fastq_text([("n", pairs[i][1], "A" * len(pairs[i][1]))])
        a1, a2 = paired_arrays(one1, one2)
        if not a1:   # empty pair: the parser yields nothing
            continue
        save(f"inline_insert_{i}", fastq1=np.frombuffer(one1, np.uint8),
             fastq2=np.frombuffer(one2, np.uint8), **insert_outputs(a1, a2))
    a1, a2 = paired_arrays(text1, text2)
    save("inline_insert_all", fastq1=np.frombuffer(text1, np.uint8),
         fastq2=np.frombuffer(text2, np.uint8), **insert_outputs(a1, a2))

    # is_mate (tests/test_fastq_record_array.py:22-43)
    names = [("same", "same", True), ("same1", "same2", True),
             ("same with comments", "same different comments", True),
             ("same1 with comments", "same2 different comments", True),
             ("same1", "same2 with comments", True), ("same1 with comments", "same2", True),
             ("differnt", "diferent", False), ("different with comment", "diferent with comment", False),
             ("same1", "same3", False), ("same2", "same1", True), ("same2", "same5", False)]
    got = []
    for a, b, _ in names:
        x = _qc.FastqRecordArrayView([_qc.FastqRecordView(a, "A", "A")])
        y = _qc.FastqRecordArrayView([_qc.FastqRecordView(b, "A", "A")])
        got.append(x.is_mate(y))
    with open(os.path.join(HERE, "is_mate.json"), "wt") as f:
        json.dump([[a, b, bool(g)] for (a, b, _), g in zip(names, got)], f, indent=0)


def invalid_phred():
    """The state the reference's QCMetrics is left in behind `ValueError: Not a valid phred
    character` (_qcmodule.c:2073-2075, 2102-2105), and that the object stays usable: arrays
    before, the array that raises (bad byte in the unrolled part, in the trailing 1-4 qualities,
    at position 0, two bad reads in one array, a longer read behind the offender), one array
    after it.  -> inline_qc_invalid_phred_<k>.npz"""
    rng = np.random.default_rng(99)

    def fastq(n, L, bad=()):
        recs = []
        for i in range(n):
            Li = L if isinstance(L, int) else int(L[i])
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=Li, p=[.24, .24, .24, .24, .04]).tobytes()
            q = bytearray((rng.integers(0, 94, size=Li) + 33).astype(np.uint8).tobytes())
            for (r, pos, ch) in bad:
                if r == i:
                    q[pos] = ch
            recs.append(b"@r%d\n%s\n+\n%s\n" % (i, s, bytes(q)))
        return b"".join(recs)

    cases = [
        (150, [(70, 149, 32)]),            # the trailing qualities (:2100-2105)
        (150, [(3, 0, 127)]),              # position 0, a byte above the range
        (100, [(191, 64, 10 + 22)]),       # inside the unrolled part (:2073-2075)
        (70, [(130, 67, 32), (150, 3, 32)]),   # a second bad read behind the first: never looked at
        (np.r_[np.full(100, 40), 300, np.full(99, 40)].astype(int), [(50, 39, 31)]),  # the longest read lies behind the offender
        (5, [(0, 4, 32)]),                 # the very first read of the array
    ]
    for k, (L, bad) in enumerate(cases):
        n = 200
        texts = [fastq(64, 60), fastq(n, L, bad), fastq(30, 90)]
        m = _qc.QCMetrics()
        raised = []
        errs = []
        for t in texts:
            for a in arrays_of(t, 1 << 20):
                try:
                    m.add_record_array(a)
                    raised.append("")
                except ValueError as e:
                    raised.append(str(e))
                errs.append(error_rates_of(a))
        save(f"inline_qc_invalid_phred_{k}", fastq0=np.frombuffer(texts[0], np.uint8), fastq1=np.frombuffer(texts[1], np.uint8),
             fastq2=np.frombuffer(texts[2], np.uint8), raised=np.array(raised),
             qc_base=np.array(m.base_count_table(), np.uint64), qc_phred=np.array(m.phred_count_table(), np.uint64),
             qc_ea_base=np.array(m.end_anchored_base_count_table(), np.uint64),
             qc_ea_phred=np.array(m.end_anchored_phred_count_table(), np.uint64),
             qc_gc=np.array(m.gc_content(), np.uint64), qc_phred_scores=np.array(m.phred_scores(), np.uint64),
             qc_number_of_reads=np.uint64(m.number_of_reads), qc_max_length=np.uint64(m.max_length),
             qc_error_rates=np.concatenate(errs))


def synthetic():
    """(3) slices of the build's own synthetic generator (needs libsqgpu.so's host
    generator; run after `python -c 'import __graft_entry__ as g; g.build()'`)."""
    from sequali_amd import synth
    text = synth.illumina_fastq(0, 20000, seed=synth.DEFAULT_SEED)
    single_end_case("synth_illumina_20k", text, ILLUMINA_PROBES, gen=(synth.ILLUMINA, 0, 20000, synth.DEFAULT_SEED))
    t1, t2 = synth.illumina_paired_fastq(0, 20000, seed=synth.DEFAULT_SEED)
    a1, a2 = paired_arrays(t1, t2)
    out = text_entry("fastq1", t1, (synth.ILLUMINA, 0, 20000, synth.DEFAULT_SEED))
    out.update(text_entry("fastq2", t2, (synth.ILLUMINA_R2, 0, 20000, synth.DEFAULT_SEED)))
    out.update(qc_outputs(a1, prefix="qc1_"))
    out.update(qc_outputs(a2, prefix="qc2_"))
    out.update(pertile_outputs(a1, prefix="pt1_"))
    out.update(pertile_outputs(a2, prefix="pt2_"))
    out.update(dedup_outputs(a1, a2, prefix="dd_", front_sequence_offset=0, back_sequence_offset=0))
    out.update(dedup_outputs(a1, a2, prefix="ddcap_", max_stored_fingerprints=1000,
                             front_sequence_offset=0, back_sequence_offset=0))
    out.update(insert_outputs(a1, a2))
    out.update(insert_outputs(a1, a2, prefix="iscap_", max_adapters=16))
    save("synth_illumina_paired_20k", **out)
    text = synth.nanopore_fastq(0, 300, seed=synth.DEFAULT_SEED)
    single_end_case("synth_nanopore_300", text, NANOPORE_PROBES, gen=(synth.NANOPORE, 0, 300, synth.DEFAULT_SEED))


def metas_of(arr) -> np.ndarray:
    """The FastqMeta structs of a reference array object as (n, 7) integers with
    record_start made relative to the array's bytes object."""
    n = len(arr)
    raw = bytes((ctypes.c_char * (n * 40)).from_address(id(arr) + 32))
    dt = np.dtype([("record_start", "<u8"), ("name_length", "<u4"), ("sequence_offset", "<u4"),
                   ("sequence_length", "<u4"), ("qualities_offset", "<u4"), ("tags_offset", "<u4"),
                   ("tags_length", "<u4"), ("err", "<f8")])
    m = np.frombuffer(raw, dtype=dt)
    base = ctypes.cast(ctypes.c_char_p(arr.obj), ctypes.c_void_p).value
    out = np.zeros((n, 7), dtype=np.int64)
    out[:, 0] = m["record_start"].astype(np.int64) - base
    for k, f in enumerate(dt.names[1:7], 1):
        out[:, k] = m[f]
    return out


def parser():
    """(4) what the reference's FastqParser does with a text at a buffer size: the
    arrays it yields (bytes object, FastqMeta structs) or the exception it raises.
    -> parser_cases.npz / parser_errors.json, the oracle of the GPU record split."""
    good = {
        "simple": read_file("simple.fastq"),
        "illumina100": read_file("100_illumina_adapters.fastq"),
        "nanopore100": read_file("100_nanopore_reads.fastq.gz"),
        "crlf": b"@r1\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nAC\r\n+\r\nII\r\n",
        "empty_lines": b"@\n\n+\n\n@x\n\n+\n\n@y y\nA\n+y\n!\n",
        "at_in_quals": b"@r1\nACGT\n+\n@@@@\n@r2\nAC\n+r2\n+@\n@r3\n@\n+\n+\n",
        "empty": b"",
    }
    out = {}
    names = []
    for name, text in good.items():
        for bs in (1, 7, 64, 300, 4096, 128 * 1024):
            if bs < 64 and len(text) > 50_000:
                continue
            arrays = arrays_of(text, bs)
            key = f"{name}_{bs}"
            names.append(key)
            out[name + "_text"] = np.frombuffer(text, np.uint8)
            out[key + "_buffersize"] = np.int64(bs)
            out[key + "_sizes"] = np.array([len(a) for a in arrays], np.int64)
            out[key + "_objlens"] = np.array([len(a.obj) for a in arrays], np.int64)
            out[key + "_metas"] = (np.concatenate([metas_of(a) for a in arrays])
                                   if arrays else np.zeros((0, 7), np.int64))
    out["names"] = np.array(names)
    save("parser_cases", **out)

    rec = b"@SOMEHEADER METADATA MOREMETADATA\nAGA\n+\nGGG\n"
    body = b"".join(b"@r%d\nACGTACGT\n+\nIIIIIIII\n" % i for i in range(40))
    bad = [b"not a record", b"@correctname\nSEQ\n-\n", b"@correctname\nAGA\n+\nGG\n",
           "@n\u00c4m\u00e9 \nAGC\n+\nHHH\n".encode("latin-1"),
           body + b"r40\nAC\n+\nII\n" + body,
           body + b"@r40\nAC\n-\nII\n" + body,
           body + b"@r40\nACG\n+\nII\n" + body,
           body + b"@r40\nACG\n+\nII\n" + b"x41\nAC\n-\nI\n" + body,       # first error wins
           body + b"@r40\nAC\n-\nIII\n" + body,                                # '+' before lengths
           body + b"\n" + body,                                                 # blank line
           body + b"@tail\nAC\nX",                                              # bad '+' in the tail
           body + b"tail",                                                      # bad '@' in the tail
           body + b"ta",                                                        # too short to look at
           body + b"@tail\nAC\n",                                               # tail ends before '+'
           body + "@r\nAC\n+\nI\u00ff\n".encode("latin-1") + body,
           body + b"x\nAC\n+\nII\n" + "\u00e9".encode("latin-1")]           # ASCII check first
    bad += [rec[:end] for end in range(1, len(rec))]
    errors = []
    for text in bad:
        for bs in (8, 100, 128 * 1024):
            try:
                arrays = arrays_of(text, bs)
                res = {"sizes": [len(a) for a in arrays]}
            except Exception as e:  # noqa: BLE001
                res = {"error": type(e).__name__, "message": str(e)}
            errors.append({"text": text.decode("latin-1"), "buffersize": bs, **res})
    with open(os.path.join(HERE, "parser_errors.json"), "wt") as f:
        json.dump(errors, f, indent=0)


def full_metas_of(arr) -> np.ndarray:
    """FastqMeta structs (40 bytes) of a reference array with record_start made an offset"""
    n = len(arr)
    raw = bytearray((ctypes.c_char * (n * 40)).from_address(id(arr) + 32))
    m = np.frombuffer(raw, dtype=np.dtype([("record_start", "<u8"), ("rest", "V24"), ("err", "<f8")]))
    base = ctypes.cast(ctypes.c_char_p(arr.obj), ctypes.c_void_p).value
    inside = (m["record_start"] >= base) & (m["record_start"] < base + max(len(arr.obj), 1))
    if n and inside.all():
        m["record_start"] -= np.uint64(base)
    else:
        # an array built from views keeps pointing at the views' own buffers (:672-684);
        # its obj is the same records back to back: name|sequence|qualities|tags
        f = np.frombuffer(bytes(raw), dtype=np.dtype([("p", "<u8"), ("nl", "<u4"), ("so", "<u4"), ("sl", "<u4"),
                                                      ("qo", "<u4"), ("to", "<u4"), ("tl", "<u4"), ("e", "<f8")]))
        sizes = f["nl"].astype(np.uint64) + 2 * f["sl"].astype(np.uint64) + f["tl"].astype(np.uint64)
        m["record_start"] = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64) if n else m["record_start"]
    return np.frombuffer(bytes(raw), dtype=np.uint8).reshape(n, 40)


def nanostats():
    """(5) NanoStats (_qcmodule.c:4804-5430) on the reference's nanopore data, on its BAM
    file through the reference's BamParser (the arrays are stored, so the tags path is
    pinned without a BAM parser on the other side) and on inline tag / header cases.
    QCMetrics runs first, as in the driver (__main__.py:279-306), so that the metas carry
    accumulated_error_rate.  -> nanostats_cases.npz"""
    import struct
    import warnings
    cases = {}

    def arrays_from_views(views, per_array=3):
        return [_qc.FastqRecordArrayView(views[i:i + per_array]) for i in range(0, len(views), per_array)]

    def z(b: bytes) -> bytes:
        return b + b"\x00"

    def tag(name: bytes, typ: bytes, payload: bytes) -> bytes:
        return name + typ + payload

    uuid = b"8D8AC610-566D-4EF0-9C22-186B2A5ED793"
    good_tags = (tag(b"rn", b"C", struct.pack("<B", 10)) + tag(b"ch", b"S", struct.pack("<H", 444)) +
                 tag(b"st", b"Z", z(b"2021-09-30T11:34:08Z")) + tag(b"RG", b"Z", z(b"SS_A1")) +
                 tag(b"du", b"f", struct.pack("<f", 2.5)) + tag(b"pi", b"Z", z(uuid)))
    V = _qc.FastqRecordView
    hdr = "cb1dab45-aa4c-43fc-a91e-ad0ecc92f5c9 runid=c989 read=10 ch=%d start_time=%s flow_cell_id=PAI09842"
    header_views = [
        V(hdr % (444, "2021-09-30T11:34:08Z"), "ACGT", "AAAA"),
        V(hdr % (1, "2019-01-26T18:52:46.123456+02:00"), "ACGTA", "AAAA5"),
        V(hdr % (2048, "2024-02-29T23:59:59-09:30"), "A", "!"),
        V(hdr % (12, "1970-01-01T00:00:00Z"), "", ""),
        V("x ch=5 start_time=2021-09-30T11:34:08Z", "ACGT", "IIII"),
        V("x start_time=2021-09-30T11:34:09Z ch=000000000000000018", "ACGT", "IIII"),
        V("x ch=3 start_time=2038-01-19T03:14:08Z extra=a=b", "ACGT", "IIII"),
        V("x ch=3 start_time=2100-03-01T00:00:00Z", "ACGT", "IIII"),
    ]
    cases["headers_ok"] = arrays_from_views(header_views)
    for k, bad in enumerate(["noseparator", "x ch=5", "x start_time=2021-09-30T11:34:08Z", "x ch=5 novalue",
                             "x ch=-1 start_time=2021-09-30T11:34:08Z", "x ch=5 start_time=2021-09-30 11:34:08Z",
                             "x ch=5 start_time=1969-09-30T11:34:08Z", "x ch=5 start_time=2021-13-30T11:34:08Z",
                             "x ch=5 start_time=2021-09-30T11:34:08", "x ch=5 start_time=2021-09-30T11:34:08+0200",
                             "x ch=1234567890123456789 start_time=2021-09-30T11:34:08Z",
                             "@SIM:1:FCX:1:1101:1:1 1:N:0:ATCCGA"]):
        cases[f"header_bad_{k}"] = arrays_from_views(header_views[:4] + [V(bad, "ACGT", "IIII")] + header_views[4:])
    tag_views = [
        V("r1", "ACGT", "AAAA", good_tags),
        V("r2", "ACGTAC", "AAAAAA", tag(b"ch", b"c", struct.pack("<b", -3)) + tag(b"st", b"Z", z(b"2022-05-06T07:08:09.5+01:00"))),
        V("r3", "AC", "AA", tag(b"ch", b"I", struct.pack("<I", 70000)) + tag(b"du", b"f", struct.pack("<f", 0.125))),
        V("r4", "ACG", "AAA", tag(b"ch", b"i", struct.pack("<i", 9)) + tag(b"st", b"Z", z(b"2022-05-06T07:08:10Z")) +
          tag(b"XA", b"A", b"q") + tag(b"XB", b"B", b"s" + struct.pack("<I3h", 3, 1, 2, 3)) +
          tag(b"XH", b"H", z(b"1AE301")) + tag(b"pi", b"Z", z(uuid.lower()))),
        V("r5", "A", "A", tag(b"st", b"Z", z(b"garbage")) + tag(b"ch", b"s", struct.pack("<h", 77))),
        V("r6", "ACGT", "AAAA", tag(b"pi", b"Z", z(b"8D8AC610-566D-3EF0-9C22-186B2A5ED793")) + tag(b"ch", b"C", b"\x05")),
        V("r7", "ACGT", "AAAA", tag(b"pi", b"Z", z(b"8D8AC61G-566D-4EF0-9C22-186B2A5ED793")) + tag(b"ch", b"C", b"\x06")),
        V("r8", "ACGT", "AAAA", tag(b"ch", b"C", b"\x07") + tag(b"st", b"Z", z(b"2023-01-01T00:00:00Z"))),
    ]
    cases["tags_ok"] = arrays_from_views(tag_views)
    cases["tags_pi_short"] = arrays_from_views(tag_views[:2] + [V("w", "AC", "AA", tag(b"pi", b"Z", z(b"tooshort")) + tag(b"ch", b"C", b"\x01"))] + tag_views[2:])
    bad_tags = [b"ch", tag(b"ch", b"S", b"\x01"), tag(b"st", b"Z", b"2021-09-30T11:34:08Z"),
                tag(b"XB", b"B", b"Z" + struct.pack("<I", 1) + b"a\x00"), tag(b"XQ", b"Q", b"1234"),
                tag(b"st", b"A", b"x"), tag(b"du", b"i", struct.pack("<i", 5)), tag(b"pi", b"i", struct.pack("<i", 5)),
                tag(b"ch", b"f", struct.pack("<f", 1.0)), tag(b"ch", b"Z", z(b"12")),
                tag(b"XB", b"B", b"i" + struct.pack("<I", 5) + b"\x00" * 8), tag(b"XB", b"B", b"c\x01")]
    for k, t in enumerate(bad_tags):
        cases[f"tags_bad_{k}"] = arrays_from_views(tag_views[:4] + [V("bad", "ACGT", "AAAA", t)] + tag_views[4:])
    for name in ("100_nanopore_reads.fastq.gz", "nanopore_disparate_dates.fastq", "single_nanopore_metadata.fastq",
                 "empty_nanopore_metadata.fastq", "100_illumina_adapters.fastq", "empty.fastq"):
        cases["file_" + name.split(".")[0]] = arrays_of(read_file(name), 16 * 1024)
    with open(os.path.join(REFDATA, "dorado_nanopore_100reads.bam"), "rb") as f:
        bam = gzip.decompress(f.read()) if False else None
    import zlib
    with open(os.path.join(REFDATA, "dorado_nanopore_100reads.bam"), "rb") as f:
        raw = f.read()
    plain, pos = b"", 0
    while pos < len(raw):  # BGZF blocks are gzip members
        d = zlib.decompressobj(31)
        plain += d.decompress(raw[pos:])
        pos = len(raw) - len(d.unused_data)
    cases["file_dorado_bam"] = list(_qc.BamParser(io.BytesIO(plain), 16 * 1024))

    out, names = {}, []
    for name, arrays in cases.items():
        qcm, ns = _qc.QCMetrics(), _qc.NanoStats()
        res = {}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                for a in arrays:
                    qcm.add_record_array(a)
                    ns.add_record_array(a)
            except BaseException as e:  # noqa: BLE001
                res = {"error": type(e).__name__, "message": str(e)}
        names.append(name)
        out[name + "_n_arrays"] = np.int64(len(arrays))
        for k, a in enumerate(arrays):
            out[f"{name}_obj{k}"] = np.frombuffer(a.obj, np.uint8)
            out[f"{name}_metas{k}"] = full_metas_of(a)
        infos = list(ns.nano_info_iterator())
        out[name + "_infos"] = np.array([(i.start_time, i.duration, i.channel_id, i.length, 0,
                                          i.cumulative_error_rate, i.parent_id_hash) for i in infos],
                                        dtype=[("start_time", "<i8"), ("duration", "<f4"), ("channel_id", "<i4"),
                                               ("length", "<u4"), ("pad", "<u4"), ("cumulative_error_rate", "<f8"),
                                               ("parent_id_hash", "<u8")])
        out[name + "_scalars"] = np.array([ns.number_of_reads, ns.minimum_time, ns.maximum_time], np.int64)
        out[name + "_result"] = np.array(json.dumps({"skipped_reason": ns.skipped_reason, **res,
                                                      "warnings": [str(w.message) for w in caught]}))
        print(name, len(infos), ns.skipped_reason, res, [str(w.message)[:40] for w in caught])
    out["names"] = np.array(names)
    save("nanostats_cases", **out)


def bgzf_plain(name: str) -> bytes:
    import zlib
    with open(os.path.join(REFDATA, name), "rb") as f:
        raw = f.read()
    if not raw.startswith(b"\x1f\x8b"):
        return raw
    plain, pos = b"", 0
    while pos < len(raw):  # BGZF blocks are gzip members
        d = zlib.decompressobj(31)
        plain += d.decompress(raw[pos:])
        pos = len(raw) - len(d.unused_data)
    return plain


def bam():
    """(6) what the reference's BamParser (_qcmodule.c:1386-1703) does with an uncompressed
    BAM stream at a buffer size: header, the arrays it yields (record count, FastqMeta
    structs, the used part of the decoded buffer) or the exception.  The BAM streams are
    the reference's test files, BGZF removed.  -> bam_cases.npz, bam_errors.json"""
    files = ["simple.unaligned.bam", "simple.raw.bam", "missing_quals.bam", "test_skip.bam",
             "project.NIST_NIST7035_H7AP8ADXX_TAAGGCGA_1_NA12878.bwa.markDuplicates.bam",
             "dorado_nanopore_100reads.bam", "secondary_alignment.bam"]
    out, names = {}, []
    for fname in files:
        plain = bgzf_plain(fname)
        key = fname[:-4].replace(".", "_")[:40]
        out[key + "_bam"] = np.frombuffer(plain, np.uint8)
        for bs in (4, 10, 40, 1000, 48 * 1024, 1 << 20):
            if bs < 1000 and len(plain) > 100_000:
                continue
            parser = _qc.BamParser(io.BytesIO(plain), bs)
            arrays = list(parser)
            k = f"{key}_{bs}"
            names.append(k)
            out[k + "_header"] = np.frombuffer(parser.header, np.uint8)
            out[k + "_sizes"] = np.array([len(a) for a in arrays], np.int64)
            metas = [metas_of(a) for a in arrays]
            out[k + "_metas"] = np.concatenate(metas) if metas else np.zeros((0, 7), np.int64)
            used = []
            for a, m in zip(arrays, metas):
                end = int((m[:, 0] + m[:, 5] + m[:, 6]).max()) if len(m) else 0  # start + tags_offset + tags_length
                used.append(np.frombuffer(a.obj[:end], np.uint8))
            out[k + "_used"] = np.concatenate(used) if used else np.zeros(0, np.uint8)
            out[k + "_used_lens"] = np.array([len(u) for u in used], np.int64)
            print(k, [len(a) for a in arrays][:8], len(arrays))
    out["names"] = np.array(names)
    save("bam_cases", **out)

    raw = bgzf_plain("simple.raw.bam")
    record_with_header, header = raw[:115], raw[:54]
    errors = []
    cases = [record_with_header[:end] for end in range(0, len(record_with_header))]
    cases += [b"@my header", raw[:200], raw + raw[54:115][:30]]
    for data in cases:
        for bs in (4, 48 * 1024):
            try:
                arrays = list(_qc.BamParser(io.BytesIO(data), bs))
                res = {"sizes": [len(a) for a in arrays]}
            except Exception as e:  # noqa: BLE001
                res = {"error": type(e).__name__, "message": str(e)}
            errors.append({"data": data.decode("latin-1"), "buffersize": bs, **res})
    for bs in (0, 3):
        try:
            _qc.BamParser(io.BytesIO(raw), bs)
        except Exception as e:  # noqa: BLE001
            errors.append({"data": raw.decode("latin-1"), "buffersize": bs, "error": type(e).__name__, "message": str(e)})
    with open(os.path.join(HERE, "bam_errors.json"), "wt") as f:
        json.dump(errors, f, indent=0)
    print(len(errors), "error cases;", sorted({e.get("error", "ok") for e in errors}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bam":
        bam()
    elif len(sys.argv) > 1 and sys.argv[1] == "nanostats":
        nanostats()
    elif len(sys.argv) > 1 and sys.argv[1] == "invalid_phred":
        invalid_phred()
    elif len(sys.argv) > 1 and sys.argv[1] == "synthetic":
        synthetic()
    elif len(sys.argv) > 1 and sys.argv[1] == "parser":
        parser()
    else:
        main()
