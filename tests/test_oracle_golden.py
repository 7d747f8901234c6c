"""Pins the CPU oracle (oracle/sq_oracle.c) against golden vectors captured
from the compiled reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import golden, golden_json, golden_names, golden_text, kwargs_of, split_fastq


def check_qc(g, buf, metas, prefix="qc_"):
    m = oracle.QCMetrics(int(g[prefix + "end_anchor"]))
    metas = metas.copy()
    m.add(buf, metas)
    assert m.number_of_reads == int(g[prefix + "number_of_reads"])
    assert m.max_length == int(g[prefix + "max_length"])
    for key, got in [("base", m.base_count_table()), ("phred", m.phred_count_table()),
                     ("ea_base", m.end_anchored_base_count_table()),
                     ("ea_phred", m.end_anchored_phred_count_table()),
                     ("gc", m.gc_content()), ("phred_scores", m.phred_scores())]:
        np.testing.assert_array_equal(got, g[prefix + key], err_msg=key)
    # bit-exact, not approximately equal
    np.testing.assert_array_equal(metas["accumulated_error_rate"].view(np.uint64),
                                  g[prefix + "error_rates"].view(np.uint64))


def check_adapter(g, buf, metas, prefix="ad_"):
    c = oracle.AdapterCounter([str(p) for p in g[prefix + "probes"]])
    c.add(buf, metas)
    assert c.max_length == int(g[prefix + "max_length"])
    assert c.number_of_sequences == int(g[prefix + "number_of_sequences"])
    for i, (_, f, r) in enumerate(c.get_counts()):
        np.testing.assert_array_equal(f, g[prefix + "fwd"][i])
        np.testing.assert_array_equal(r, g[prefix + "rev"][i])


def check_pertile(g, buf, metas, prefix="pt_"):
    p = oracle.PerTileQuality()
    p.add(buf, metas)
    assert p.max_length == int(g[prefix + "max_length"])
    assert p.number_of_reads == int(g[prefix + "number_of_reads"])
    reason = str(g[prefix + "skipped_reason"])
    assert p.skipped == bool(reason)
    if reason:
        m = metas[p.skipped_record]
        name = bytes(buf[int(m["record_start"]):int(m["record_start"]) + int(m["name_length"])]).decode()
        assert name in reason
    tc = p.get_tile_counts()
    np.testing.assert_array_equal([t for t, _, _ in tc], g[prefix + "tiles"])
    for i, (_, e, c) in enumerate(tc):
        # the oracle keeps the reference's summation order: exact
        np.testing.assert_array_equal(e.view(np.uint64), g[prefix + "errors"][i].view(np.uint64))
        np.testing.assert_array_equal(c, g[prefix + "counts"][i])


def check_overrep(g, buf, metas, prefix="ov_"):
    o = oracle.OverrepresentedSequences(**kwargs_of(g, prefix))
    o.add(buf, metas)
    for k in ("number_of_sequences", "sampled_sequences", "total_fragments",
              "collected_unique_fragments"):
        assert getattr(o, k) == int(g[prefix + k]), k
    sc = o.sequence_counts()
    assert sc == {str(s): int(c) for s, c in zip(g[prefix + "seqs"], g[prefix + "counts"])}
    ovr = o.overrepresented_sequences()
    assert [c for c, _, _ in ovr] == [int(x) for x in g[prefix + "ovr_counts"]]
    assert [s for _, _, s in ovr] == [str(x) for x in g[prefix + "ovr_seqs"]]
    assert [f for _, f, _ in ovr] == [float(x) for x in g[prefix + "ovr_fracs"]]


def check_dedup(g, batch1, batch2=None, prefix="dd_"):
    d = oracle.DedupEstimator(**kwargs_of(g, prefix))
    if batch2 is None:
        d.add(*batch1)
    else:
        d.add_pair(*batch1, *batch2)
    assert d._modulo_bits == int(g[prefix + "modulo_bits"])
    assert d.tracked_sequences == int(g[prefix + "tracked_sequences"])
    assert d._hash_table_size == int(g[prefix + "hash_table_size"])
    # slot order too: the oracle reproduces the table layout
    np.testing.assert_array_equal(d.duplication_counts(), g[prefix + "counts_slot_order"])


def check_insert(g, batch1, batch2, prefix="is_"):
    z = oracle.InsertSizeMetrics(**kwargs_of(g, prefix))
    z.add_pair(*batch1, *batch2)
    np.testing.assert_array_equal(z.insert_sizes(), g[prefix + "insert_sizes"])
    assert z.total_reads == int(g[prefix + "total_reads"])
    assert z.number_of_adapters_read1 == int(g[prefix + "n_adapters_read1"])
    assert z.number_of_adapters_read2 == int(g[prefix + "n_adapters_read2"])
    for which, got in (("ad1", z.adapters_read1()), ("ad2", z.adapters_read2())):
        want = list(zip([str(s) for s in g[prefix + which + "_seqs"]],
                        [int(c) for c in g[prefix + which + "_counts"]]))
        assert got == want  # slot order included


def test_error_table_bits():
    want = [int(x, 16) for x in golden_json("error_table")]
    got = [int(np.float64(oracle.error_rate(q)).view(np.uint64)) for q in range(94)]
    assert got == want


@pytest.mark.parametrize("name", golden_names("ref_[!L]*") + golden_names("synth_*[!d]_[0-9]*"))
def test_single_end_files(name):
    g = golden(name)
    buf, metas = split_fastq(golden_text(g, "fastq"))
    check_qc(g, buf, metas)
    check_adapter(g, buf, metas)
    check_pertile(g, buf, metas)
    for p in ("ov_", "ov1_", "ovcap_"):
        check_overrep(g, buf, metas, p)
    for p in ("dd_", "dd0_", "ddcap_"):
        check_dedup(g, (buf, metas), prefix=p)


@pytest.mark.parametrize("name", ["ref_LTB_paired"] + golden_names("synth_*paired*"))
def test_paired_files(name):
    g = golden(name)
    b1 = split_fastq(golden_text(g, "fastq1"))
    b2 = split_fastq(golden_text(g, "fastq2"))
    check_qc(g, *b1, prefix="qc1_")
    check_qc(g, *b2, prefix="qc2_")
    check_pertile(g, *b1, prefix="pt1_")
    check_pertile(g, *b2, prefix="pt2_")
    for p in ("dd_", "ddcap_"):
        check_dedup(g, b1, b2, prefix=p)
    for p in ("is_", "iscap_"):
        check_insert(g, b1, b2, prefix=p)
    if "ov1_kwargs" in g:
        check_overrep(g, *b1, prefix="ov1_")
        check_overrep(g, *b2, prefix="ov2_")
    if "ad1_probes" in g:
        check_adapter(g, *b1, prefix="ad1_")


@pytest.mark.parametrize("name", golden_names("inline_qc_[0-9]*"))
def test_inline_qc(name):
    g = golden(name)
    check_qc(g, *split_fastq(golden_text(g, "fastq")))


def test_inline_qc_long_quality():
    g = golden("inline_qc_long_quality")
    n = int(g["n"])
    text = b"@name\n" + b"A" * n + b"\n+\n" + b"!" * 1000 + b"S" * (n - 1000) + b"\n"
    buf, metas = split_fastq(text)
    m = oracle.QCMetrics()
    m.add(buf, metas)
    np.testing.assert_array_equal(m.phred_scores(), g["qc_phred_scores"])
    np.testing.assert_array_equal(m.gc_content(), g["qc_gc"])
    assert metas["accumulated_error_rate"][0] == g["qc_error_rates"][0]


def test_h1_uniform_quality_grid():
    g = golden("h1_uniform_quality_grid")
    for q in range(94):
        for j, L in enumerate(g["lengths"]):
            L = int(L)
            buf, metas = oracle.make_batch(["n"], ["A" * L], [chr(q + 33) * L])
            m = oracle.QCMetrics()
            m.add(buf, metas)
            assert int(np.nonzero(m.phred_scores())[0][0]) == g["bins"][q, j], (q, L)
            assert metas["accumulated_error_rate"][0] == g["error_rates"][q, j]


@pytest.mark.parametrize("name", golden_names("inline_adapter_*"))
def test_inline_adapter(name):
    g = golden(name)
    check_adapter(g, *split_fastq(golden_text(g, "fastq")))


@pytest.mark.parametrize("name", golden_names("inline_pertile_*"))
def test_inline_pertile(name):
    g = golden(name)
    check_pertile(g, *split_fastq(golden_text(g, "fastq")))


@pytest.mark.parametrize("name", golden_names("inline_overrep_*"))
def test_inline_overrep(name):
    g = golden(name)
    check_overrep(g, *split_fastq(golden_text(g, "fastq")))


@pytest.mark.parametrize("name", golden_names("inline_dedup_cap*"))
def test_inline_dedup_caps(name):
    g = golden(name)
    check_dedup(g, split_fastq(golden_text(g, "fastq")))


@pytest.mark.parametrize("name", golden_names("inline_dedup_geom_*"))
def test_inline_dedup_geometry(name):
    g = golden(name)
    check_dedup(g, split_fastq(golden_text(g, "fastq")))
    check_dedup(g, split_fastq(golden_text(g, "fastq1")), split_fastq(golden_text(g, "fastq2")), prefix="ddp_")


@pytest.mark.parametrize("name", golden_names("inline_insert_*"))
def test_inline_insert(name):
    g = golden(name)
    check_insert(g, split_fastq(golden_text(g, "fastq1")), split_fastq(golden_text(g, "fastq2")))


def test_is_mate():
    for a, b, want in golden_json("is_mate"):
        assert oracle.names_are_mates(a, b) is want, (a, b)


@pytest.mark.parametrize("name", golden_names("inline_qc_invalid_phred_*"))
def test_qc_state_behind_an_invalid_phred_character(name):
    """_qcmodule.c:2073-2075, 2102-2105: what the reference has counted when it raises, and that
    it goes on counting afterwards"""
    g = golden(name)
    m = oracle.QCMetrics()
    errs = []
    for k, msg in enumerate(g["raised"]):
        buf, metas = split_fastq(golden_text(g, f"fastq{k}"))
        if msg:
            with pytest.raises(ValueError) as e:
                m.add(buf, metas)
            assert str(e.value) == str(msg)
        else:
            m.add(buf, metas)
        errs.append(metas["accumulated_error_rate"])
    assert m.number_of_reads == int(g["qc_number_of_reads"]) and m.max_length == int(g["qc_max_length"])
    for key, got in [("base", m.base_count_table()), ("phred", m.phred_count_table()),
                     ("ea_base", m.end_anchored_base_count_table()), ("ea_phred", m.end_anchored_phred_count_table()),
                     ("gc", m.gc_content()), ("phred_scores", m.phred_scores())]:
        np.testing.assert_array_equal(got, g["qc_" + key], err_msg=key)
    np.testing.assert_array_equal(np.concatenate(errs).view(np.uint64), g["qc_error_rates"].view(np.uint64))
