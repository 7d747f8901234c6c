"""The product against the reference ITSELF (oracle/_ref travels to the GPU box; the reference's sources do not): the same
FASTQ text through sequali_amd.FastqParser + the HIP modules and through the reference's FastqParser + its C modules, random
reads, random module parameters, three buffer sizes -- every getter of SURVEY 8a, bit for bit (PerTileQuality's f64 sums
to 1e-6: the only sums whose order differs).  The other GPU tests go through the oracle; this one has no middle man."""
import io
import warnings

import numpy as np
import pytest

from tests.test_oracle_vs_reference import ADAPTER_SETS, REF, draw, fastq, u64

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")]


@pytest.mark.parametrize("seed", range(30))
def test_single_end_modules(seed):
    import sequali_amd as S
    rng = np.random.default_rng(51000 + seed)
    n = int(rng.choice([1, 17, 64, 400, 3000, 20000]))
    max_len = int(rng.choice([5, 40, 151, 300, 1200]))
    if n * max_len > 3_000_000:
        n = 3_000_000 // max_len
    adapters = ADAPTER_SETS[int(rng.integers(0, len(ADAPTER_SETS)))]
    kind = str(rng.choice(["illumina", "illumina", "breaks", "plain"]))
    names, seqs, quals = draw(rng, n, max_len, bool(rng.random() < 0.4), kind, adapters)
    text = fastq(names, seqs, quals)
    buffer_size = int(rng.choice([4096, 128 * 1024, 1 << 24]))
    ea = int(rng.choice([0, 7, 100, 300]))
    okw = dict(max_unique_fragments=int(rng.choice([50, 700, 5_000_000])), sample_every=int(rng.choice([1, 3, 8])),
               fragment_length=int(rng.choice([5, 21, 31])))
    dkw = dict(max_stored_fingerprints=int(rng.choice([100, 300, 1_000_000])),
               front_sequence_offset=int(rng.choice([0, 8, 64])), back_sequence_offset=int(rng.choice([0, 8, 64])))
    fused = bool(rng.random() < 0.5)
    res = {}
    for M in (REF, S):
        mods = dict(q=M.QCMetrics(ea), a=M.AdapterCounter(adapters), p=M.PerTileQuality(),
                    o=M.OverrepresentedSequences(**okw), d=M.DedupEstimator(**dkw), n=M.NanoStats())
        one = S.FusedPass(mods["q"], mods["a"], mods["p"]) if (M is S and fused) else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for arr in M.FastqParser(io.BytesIO(text), buffer_size):
                if one is not None:
                    one.add_record_array(arr)
                else:
                    for k in "qap":
                        mods[k].add_record_array(arr)
                for k in "odn":
                    mods[k].add_record_array(arr)
        res[M is S] = mods
    r, g = res[False], res[True]
    assert (g["q"].number_of_reads, g["q"].max_length) == (r["q"].number_of_reads, r["q"].max_length)
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        np.testing.assert_array_equal(u64(getattr(g["q"], name)()), u64(getattr(r["q"], name)()), err_msg=name)
    for (an, f, rv), (rn, fr, rr) in zip(g["a"].get_counts(), r["a"].get_counts()):
        assert an == rn
        np.testing.assert_array_equal(u64(f), u64(fr))
        np.testing.assert_array_equal(u64(rv), u64(rr))
    assert g["p"].number_of_reads == r["p"].number_of_reads and g["p"].skipped_reason == r["p"].skipped_reason
    gt, rt = g["p"].get_tile_counts(), r["p"].get_tile_counts()
    assert [t for t, _, _ in gt] == [t for t, _, _ in rt]
    for (_, e, c), (_, er, cr) in zip(gt, rt):
        np.testing.assert_allclose(np.array(e), np.array(er), rtol=1e-6, atol=0)
        np.testing.assert_array_equal(u64(c), u64(cr))
    for k in ("number_of_sequences", "sampled_sequences", "total_fragments", "collected_unique_fragments"):
        assert getattr(g["o"], k) == getattr(r["o"], k), k
    assert g["o"].sequence_counts() == r["o"].sequence_counts()
    assert g["o"].overrepresented_sequences() == r["o"].overrepresented_sequences()
    assert (g["d"]._modulo_bits, g["d"].tracked_sequences) == (r["d"]._modulo_bits, r["d"].tracked_sequences)
    np.testing.assert_array_equal(u64(g["d"].duplication_counts()), u64(r["d"].duplication_counts()))
    gi = [(i.start_time, i.channel_id, i.length, np.float64(i.cumulative_error_rate).view(np.uint64).item()) for i in g["n"].nano_info_iterator()]
    ri = [(i.start_time, i.channel_id, i.length, np.float64(i.cumulative_error_rate).view(np.uint64).item()) for i in r["n"].nano_info_iterator()]
    assert gi == ri


@pytest.mark.parametrize("seed", range(16))
def test_paired_modules(seed):
    import sequali_amd as S
    rng = np.random.default_rng(52000 + seed)
    n = int(rng.choice([1, 33, 500, 5000]))
    max_len = int(rng.choice([14, 40, 151, 260]))
    pool = []
    n1, s1, q1 = draw(rng, n, max_len, bool(rng.random() < 0.5), "illumina", None, pool)
    n2, s2, q2 = draw(rng, n, max_len, bool(rng.random() < 0.5), "illumina", None, pool)
    comp = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")
    for i in range(n):
        if rng.random() < 0.5 and len(s1[i]) >= 20:
            ins = int(rng.integers(16, len(s1[i]) + 1))
            r2 = s1[i][:ins][::-1].translate(comp) + "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
            L2 = len(s2[i])
            s2[i] = (r2 + s2[i])[:max(L2, 16)] if L2 else ""
            q2[i] = (q2[i] + "I" * len(s2[i]))[:len(s2[i])]
    s1[0] = s2[0] = "ACGTTGCAACGTTGCAAC"        # the reference's fingerprint store starts uninitialised (:4352): a long first pair defines it
    q1[0] = q2[0] = "I" * 18
    t1, t2 = fastq(n1, s1, q1), fastq(n2, s2, q2)
    dkw = dict(max_stored_fingerprints=int(rng.choice([100, 300, 1_000_000])),
               front_sequence_offset=int(rng.choice([0, 8])), back_sequence_offset=int(rng.choice([0, 8])))
    res = {}
    for M in (REF, S):
        d, z = M.DedupEstimator(**dkw), M.InsertSizeMetrics()
        a1 = list(M.FastqParser(io.BytesIO(t1), 1 << 24))
        a2 = list(M.FastqParser(io.BytesIO(t2), 1 << 24))
        d.add_record_array_pair(a1[0], a2[0])
        z.add_record_array_pair(a1[0], a2[0])
        res[M is S] = (d, z)
    (rd, rz), (gd, gz) = res[False], res[True]
    assert (gd._modulo_bits, gd.tracked_sequences) == (rd._modulo_bits, rd.tracked_sequences)
    np.testing.assert_array_equal(u64(gd.duplication_counts()), u64(rd.duplication_counts()))
    assert (gz.total_reads, gz.number_of_adapters_read1, gz.number_of_adapters_read2) == \
        (rz.total_reads, rz.number_of_adapters_read1, rz.number_of_adapters_read2)
    np.testing.assert_array_equal(u64(gz.insert_sizes()), u64(rz.insert_sizes()))
    assert list(gz.adapters_read1()) == list(rz.adapters_read1())
    assert list(gz.adapters_read2()) == list(rz.adapters_read2())
