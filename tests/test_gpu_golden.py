"""Parity of the HIP path (through the C ABI, via sequali_amd) against the golden
vectors captured from the compiled reference.  Needs a GPU."""
import io
import warnings

import numpy as np
import pytest

from tests.helpers import golden, golden_json, golden_names, golden_text, kwargs_of

pytestmark = pytest.mark.gpu


def arrays_of(text: bytes, buffersize=128 * 1024):
    from sequali_amd import FastqParser
    return list(FastqParser(io.BytesIO(text), buffersize))


def paired_arrays(text1: bytes, text2: bytes):
    from sequali_amd import FastqParser
    p1, p2 = FastqParser(io.BytesIO(text1)), FastqParser(io.BytesIO(text2))
    a1, a2 = [], []
    for arr in p1:
        a1.append(arr)
        a2.append(p2.read(len(arr)))
    return a1, a2


def u64(a):
    return np.array(a, dtype=np.uint64)


def check_qc(g, arrays, prefix="qc_", fused=None):
    from sequali_amd import QCMetrics
    m = QCMetrics(int(g[prefix + "end_anchor"]))
    for a in arrays:
        m.add_record_array(a)
    assert m.number_of_reads == int(g[prefix + "number_of_reads"])
    assert m.max_length == int(g[prefix + "max_length"])
    for key, got in [("base", m.base_count_table()), ("phred", m.phred_count_table()),
                     ("ea_base", m.end_anchored_base_count_table()),
                     ("ea_phred", m.end_anchored_phred_count_table()),
                     ("gc", m.gc_content()), ("phred_scores", m.phred_scores())]:
        np.testing.assert_array_equal(u64(got), g[prefix + key], err_msg=key)
    errs = np.concatenate([a.accumulated_error_rates() for a in arrays]) if arrays else np.zeros(0)
    np.testing.assert_array_equal(errs.view(np.uint64), g[prefix + "error_rates"].view(np.uint64))


def check_adapter(g, arrays, prefix="ad_"):
    from sequali_amd import AdapterCounter
    c = AdapterCounter([str(p) for p in g[prefix + "probes"]])
    for a in arrays:
        c.add_record_array(a)
    assert c.max_length == int(g[prefix + "max_length"])
    assert c.number_of_sequences == int(g[prefix + "number_of_sequences"])
    for i, (_, f, r) in enumerate(c.get_counts()):
        np.testing.assert_array_equal(u64(f), g[prefix + "fwd"][i], err_msg=f"fwd {i}")
        np.testing.assert_array_equal(u64(r), g[prefix + "rev"][i], err_msg=f"rev {i}")


def check_pertile(g, arrays, prefix="pt_"):
    from sequali_amd import PerTileQuality
    p = PerTileQuality()
    for a in arrays:
        p.add_record_array(a)
    assert p.max_length == int(g[prefix + "max_length"])
    assert p.number_of_reads == int(g[prefix + "number_of_reads"])
    reason = str(g[prefix + "skipped_reason"])
    assert (p.skipped_reason or "") == reason
    tc = p.get_tile_counts()
    assert [t for t, _, _ in tc] == [int(x) for x in g[prefix + "tiles"]]
    for i, (_, e, c) in enumerate(tc):
        np.testing.assert_allclose(np.array(e), g[prefix + "errors"][i], rtol=1e-6, atol=0)
        np.testing.assert_array_equal(u64(c), g[prefix + "counts"][i])


def check_overrep(g, arrays, prefix="ov_"):
    from sequali_amd import OverrepresentedSequences
    o = OverrepresentedSequences(**kwargs_of(g, prefix))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for a in arrays:
            o.add_record_array(a)
    for k in ("number_of_sequences", "sampled_sequences", "total_fragments",
              "collected_unique_fragments"):
        assert getattr(o, k) == int(g[prefix + k]), k
    assert o.sequence_counts() == {str(s): int(c) for s, c in zip(g[prefix + "seqs"], g[prefix + "counts"])}
    ovr = o.overrepresented_sequences()
    assert [c for c, _, _ in ovr] == [int(x) for x in g[prefix + "ovr_counts"]]
    assert [s for _, _, s in ovr] == [str(x) for x in g[prefix + "ovr_seqs"]]
    assert [f for _, f, _ in ovr] == [float(x) for x in g[prefix + "ovr_fracs"]]


def check_dedup(g, arrays1, arrays2=None, prefix="dd_"):
    from sequali_amd import DedupEstimator
    d = DedupEstimator(**kwargs_of(g, prefix))
    if arrays2 is None:
        for a in arrays1:
            d.add_record_array(a)
    else:
        for a, b in zip(arrays1, arrays2):
            d.add_record_array_pair(a, b)
    assert d._modulo_bits == int(g[prefix + "modulo_bits"])
    assert d.tracked_sequences == int(g[prefix + "tracked_sequences"])
    assert d._hash_table_size == int(g[prefix + "hash_table_size"])
    np.testing.assert_array_equal(u64(d.duplication_counts()), g[prefix + "counts_slot_order"])


def check_insert(g, arrays1, arrays2, prefix="is_"):
    from sequali_amd import InsertSizeMetrics
    z = InsertSizeMetrics(**kwargs_of(g, prefix))
    for a, b in zip(arrays1, arrays2):
        z.add_record_array_pair(a, b)
    np.testing.assert_array_equal(u64(z.insert_sizes()), g[prefix + "insert_sizes"])
    assert z.total_reads == int(g[prefix + "total_reads"])
    assert z.number_of_adapters_read1 == int(g[prefix + "n_adapters_read1"])
    assert z.number_of_adapters_read2 == int(g[prefix + "n_adapters_read2"])
    for which, got in (("ad1", z.adapters_read1()), ("ad2", z.adapters_read2())):
        want = list(zip([str(s) for s in g[prefix + which + "_seqs"]],
                        [int(c) for c in g[prefix + which + "_counts"]]))
        assert got == want


@pytest.mark.parametrize("name", golden_names("ref_[!L]*") + golden_names("synth_*[!d]_[0-9]*"))
def test_single_end_files(name):
    g = golden(name)
    arrays = arrays_of(golden_text(g, "fastq"))
    check_qc(g, arrays)
    check_adapter(g, arrays)
    check_pertile(g, arrays)
    for p in ("ov_", "ov1_", "ovcap_"):
        check_overrep(g, arrays, p)
    for p in ("dd_", "dd0_", "ddcap_"):
        check_dedup(g, arrays, prefix=p)


@pytest.mark.parametrize("name", golden_names("ref_100_*") + golden_names("synth_*[!d]_[0-9]*"))
def test_single_end_files_fused(name):
    """the three per-base modules in one pass give the same tables"""
    from sequali_amd import AdapterCounter, FusedPass, PerTileQuality, QCMetrics
    g = golden(name)
    arrays = arrays_of(golden_text(g, "fastq"), 64 * 1024)
    m, c, p = QCMetrics(), AdapterCounter([str(x) for x in g["ad_probes"]]), PerTileQuality()
    fused = FusedPass(m, c, p)
    for a in arrays:
        fused.add_record_array(a)
    np.testing.assert_array_equal(u64(m.base_count_table()), g["qc_base"])
    np.testing.assert_array_equal(u64(m.phred_count_table()), g["qc_phred"])
    np.testing.assert_array_equal(u64(m.end_anchored_base_count_table()), g["qc_ea_base"])
    np.testing.assert_array_equal(u64(m.end_anchored_phred_count_table()), g["qc_ea_phred"])
    np.testing.assert_array_equal(u64(m.gc_content()), g["qc_gc"])
    np.testing.assert_array_equal(u64(m.phred_scores()), g["qc_phred_scores"])
    for i, (_, f, r) in enumerate(c.get_counts()):
        np.testing.assert_array_equal(u64(f), g["ad_fwd"][i])
        np.testing.assert_array_equal(u64(r), g["ad_rev"][i])
    assert p.number_of_reads == int(g["pt_number_of_reads"])
    assert (p.skipped_reason or "") == str(g["pt_skipped_reason"])
    tc = p.get_tile_counts()
    assert [t for t, _, _ in tc] == [int(x) for x in g["pt_tiles"]]
    for i, (_, e, cnt) in enumerate(tc):
        np.testing.assert_allclose(np.array(e), g["pt_errors"][i], rtol=1e-6)
        np.testing.assert_array_equal(u64(cnt), g["pt_counts"][i])


@pytest.mark.parametrize("name", ["ref_LTB_paired"] + golden_names("synth_*paired*"))
def test_paired_files(name):
    g = golden(name)
    a1, a2 = paired_arrays(golden_text(g, "fastq1"), golden_text(g, "fastq2"))
    if "is_mate" in g:
        assert [x.is_mate(y) for x, y in zip(a1, a2)] == [bool(v) for v in g["is_mate"]]
    check_qc(g, a1, prefix="qc1_")
    check_qc(g, a2, prefix="qc2_")
    check_pertile(g, a1, prefix="pt1_")
    check_pertile(g, a2, prefix="pt2_")
    for p in ("dd_", "ddcap_"):
        check_dedup(g, a1, a2, prefix=p)
    for p in ("is_", "iscap_"):
        check_insert(g, a1, a2, prefix=p)
    if "ov1_kwargs" in g:
        check_overrep(g, a1, prefix="ov1_")
        check_overrep(g, a2, prefix="ov2_")
    if "ad1_probes" in g:
        check_adapter(g, a1, prefix="ad1_")


@pytest.mark.parametrize("name", golden_names("inline_qc_[0-9]*"))
def test_inline_qc(name):
    g = golden(name)
    check_qc(g, arrays_of(golden_text(g, "fastq")))


def test_inline_qc_long_quality():
    from sequali_amd import QCMetrics
    g = golden("inline_qc_long_quality")
    n = int(g["n"])
    text = b"@name\n" + b"A" * n + b"\n+\n" + b"!" * 1000 + b"S" * (n - 1000) + b"\n"
    arrays = arrays_of(text, 8 * 1024 * 1024)
    m = QCMetrics()
    for a in arrays:
        m.add_record_array(a)
    np.testing.assert_array_equal(u64(m.phred_scores()), g["qc_phred_scores"])
    np.testing.assert_array_equal(u64(m.gc_content()), g["qc_gc"])
    assert arrays[0].accumulated_error_rates()[0] == g["qc_error_rates"][0]


def test_h1_uniform_quality_grid():
    """658 (quality, length) combinations, all in one batch each; the phred_scores bin
    depends on the exact f64 summation order and on the host libm's log10"""
    from sequali_amd import FastqRecordArrayView, FastqRecordView, QCMetrics
    g = golden("h1_uniform_quality_grid")
    for j, L in enumerate(g["lengths"]):
        L = int(L)
        arr = FastqRecordArrayView([FastqRecordView("n", "A" * L, chr(q + 33) * L) for q in range(94)])
        m = QCMetrics()
        m.add_record_array(arr)
        want = np.zeros(94, np.uint64)
        for q in range(94):
            want[g["bins"][q, j]] += 1
        np.testing.assert_array_equal(u64(m.phred_scores()), want, err_msg=f"L={L}")
        np.testing.assert_array_equal(arr.accumulated_error_rates().view(np.uint64),
                                      g["error_rates"][:, j].copy().view(np.uint64))


@pytest.mark.parametrize("name", golden_names("inline_adapter_*"))
def test_inline_adapter(name):
    g = golden(name)
    check_adapter(g, arrays_of(golden_text(g, "fastq")))


@pytest.mark.parametrize("name", golden_names("inline_pertile_*"))
def test_inline_pertile(name):
    g = golden(name)
    check_pertile(g, arrays_of(golden_text(g, "fastq")))
    # and with the records spread over several small arrays
    check_pertile(g, arrays_of(golden_text(g, "fastq"), 300))


@pytest.mark.parametrize("name", golden_names("inline_overrep_*"))
def test_inline_overrep(name):
    g = golden(name)
    check_overrep(g, arrays_of(golden_text(g, "fastq")))
    check_overrep(g, arrays_of(golden_text(g, "fastq"), 1000))


@pytest.mark.parametrize("name", golden_names("inline_dedup_cap*"))
def test_inline_dedup_caps(name):
    g = golden(name)
    check_dedup(g, arrays_of(golden_text(g, "fastq")))
    check_dedup(g, arrays_of(golden_text(g, "fastq"), 4096))


@pytest.mark.parametrize("name", golden_names("inline_dedup_geom_*"))
def test_inline_dedup_geometry(name):
    g = golden(name)
    check_dedup(g, arrays_of(golden_text(g, "fastq")))
    a1, a2 = paired_arrays(golden_text(g, "fastq1"), golden_text(g, "fastq2"))
    check_dedup(g, a1, a2, prefix="ddp_")


@pytest.mark.parametrize("name", golden_names("inline_insert_*"))
def test_inline_insert(name):
    g = golden(name)
    a1, a2 = paired_arrays(golden_text(g, "fastq1"), golden_text(g, "fastq2"))
    check_insert(g, a1, a2)


def test_is_mate():
    from sequali_amd import FastqRecordArrayView, FastqRecordView
    for a, b, want in golden_json("is_mate"):
        x = FastqRecordArrayView([FastqRecordView(a, "A", "A")])
        y = FastqRecordArrayView([FastqRecordView(b, "A", "A")])
        assert x.is_mate(y) is want, (a, b)


@pytest.mark.parametrize("name", golden_names("inline_qc_invalid_phred_*"))
@pytest.mark.parametrize("env", [{}, {"SQ_NO_WIDE": "1", "SQ_NO_RING": "1"}, {"SQ_SPAN": "0"}])
def test_qc_state_behind_an_invalid_phred_character(name, env):
    """_qcmodule.c:2073-2075, 2102-2105.  The passes run whole batches and flag the read; the
    ValueError comes out of the next flush, which first takes back what the reference would not
    have counted (sq_qcmetrics_uncount_tail): every table, number_of_reads, max_length and the
    accumulated_error_rate of every record as in the reference behind its exception, the
    reference's message, and an object that goes on counting (the array behind the bad one).
    Once with all three arrays enqueued before the flush, once with a flush per array."""
    import os
    from sequali_amd import FastqRecordArrayView, QCMetrics
    from tests.helpers import split_fastq
    g = golden(name)
    from sequali_amd._lib import lib
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    lib().sq_knobs_reload()
    try:
        for flush_each in (False, True):
            m = QCMetrics()
            arrays, first_msg = [], None
            for k, msg in enumerate(g["raised"]):
                buf, metas = split_fastq(golden_text(g, f"fastq{k}"))
                arr = FastqRecordArrayView._from_buffer(buf, metas)
                arrays.append(arr)
                m.add_record_array(arr)
                if flush_each:
                    if msg:
                        with pytest.raises(ValueError) as e:
                            m.flush()
                        assert str(e.value) == str(msg)
                    else:
                        m.flush()
                elif msg and first_msg is None:
                    first_msg = str(msg)
            if not flush_each:
                with pytest.raises(ValueError) as e:
                    m.flush()
                assert str(e.value) == first_msg
            assert m.number_of_reads == int(g["qc_number_of_reads"]) and m.max_length == int(g["qc_max_length"])
            for key, got in [("base", m.base_count_table()), ("phred", m.phred_count_table()),
                             ("ea_base", m.end_anchored_base_count_table()), ("ea_phred", m.end_anchored_phred_count_table()),
                             ("gc", m.gc_content()), ("phred_scores", m.phred_scores())]:
                np.testing.assert_array_equal(np.array(got, np.uint64), g["qc_" + key], err_msg=key)
            errs = np.concatenate([a.accumulated_error_rates() for a in arrays])
            np.testing.assert_array_equal(errs.view(np.uint64), g["qc_error_rates"].view(np.uint64))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib().sq_knobs_reload()
