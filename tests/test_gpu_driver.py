"""The driver loop (sequali_amd.driver) on files: the reference's test data written
to disk (plain, gzip, BGZF-less BAM) goes through `run` and every module's output equals
what the oracle's modules give for the same records in the reference's order
(__main__.py:279-306)."""
import gzip
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import GOLDEN, golden, split_fastq

pytestmark = pytest.mark.gpu


def u64(a):
    return np.array(a, dtype=np.uint64)


def oracle_single(text: bytes, probes):
    buf, metas = split_fastq(text)
    q, a, p, o, d, n = (oracle.QCMetrics(), oracle.AdapterCounter(probes), oracle.PerTileQuality(),
                        oracle.OverrepresentedSequences(), oracle.DedupEstimator(front_sequence_offset=64,
                                                                                 back_sequence_offset=0),
                        oracle.NanoStats())
    q.add(buf, metas)
    a.add(buf, metas)
    p.add(buf, metas)
    o.add(buf, metas)
    d.add(buf, metas)
    n.add(buf, metas)
    return q, a, p, o, d, n


def check_qc(got, ref):
    assert got.number_of_reads == ref.number_of_reads and got.max_length == ref.max_length
    np.testing.assert_array_equal(u64(got.base_count_table()), ref.base_count_table())
    np.testing.assert_array_equal(u64(got.phred_count_table()), ref.phred_count_table())
    np.testing.assert_array_equal(u64(got.end_anchored_base_count_table()), ref.end_anchored_base_count_table())
    np.testing.assert_array_equal(u64(got.gc_content()), ref.gc_content())
    np.testing.assert_array_equal(u64(got.phred_scores()), ref.phred_scores())


@pytest.mark.parametrize("name,tech,compress", [("ref_100_illumina_adapters", "illumina", False),
                                                ("ref_100_illumina_adapters", "illumina", True),
                                                ("ref_100_nanopore", "nanopore", True),
                                                ("ref_simple", None, False)])
def test_single_end_file(tmp_path, name, tech, compress):
    import warnings
    from sequali_amd import driver
    text = golden(name)["fastq"].tobytes()
    path = tmp_path / ("reads.fastq.gz" if compress else "reads.fastq")
    path.write_bytes(gzip.compress(text) if compress else text)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = driver.run(str(path), buffersize=20_000)   # several record arrays
    assert m["sequencing_technology"] == tech
    probes = [a.sequence for a in driver.adapters_for(tech)]
    assert len(probes) == {"illumina": 6, "nanopore": 14, None: 20}[tech]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q, a, p, o, d, n = oracle_single(text, probes)
    check_qc(m["metrics"], q)
    for (s, f, r), (_, fr, rr) in zip(m["adapter_counter"].get_counts(), a.get_counts()):
        np.testing.assert_array_equal(u64(f), fr)
        np.testing.assert_array_equal(u64(r), rr)
    assert m["per_tile_quality"].number_of_reads == p.number_of_reads
    assert (m["per_tile_quality"].skipped_reason is not None) == p.skipped
    assert m["sequence_duplication"].sequence_counts() == o.sequence_counts()
    np.testing.assert_array_equal(u64(m["dedup_estimator"].duplication_counts()), d.duplication_counts())
    assert m["nanostats"].number_of_reads == n.number_of_reads
    assert (m["nanostats"].skipped_reason is not None) == n.skipped
    json.dumps(driver.raw_outputs(m))   # serialisable


def test_paired_files_and_sync_errors(tmp_path):
    from sequali_amd import driver
    g = golden("ref_LTB_paired")
    t1, t2 = g["fastq1"].tobytes(), g["fastq2"].tobytes()
    p1, p2 = tmp_path / "r1.fastq.gz", tmp_path / "r2.fastq"
    p1.write_bytes(gzip.compress(t1))
    p2.write_bytes(t2)
    m = driver.run(str(p1), str(p2), buffersize=50_000)
    assert m["sequencing_technology"] == "illumina" and m["adapter_counter"] is None
    b1, m1 = split_fastq(t1)
    b2, m2 = split_fastq(t2)
    q1, q2, z = oracle.QCMetrics(), oracle.QCMetrics(), oracle.InsertSizeMetrics()
    d = oracle.DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
    q1.add(b1, m1)
    q2.add(b2, m2)
    z.add_pair(b1, m1, b2, m2)
    d.add_pair(b1, m1, b2, m2)
    check_qc(m["metrics"], q1)
    check_qc(m["metrics_reverse"], q2)
    np.testing.assert_array_equal(u64(m["insert_size_metrics"].insert_sizes()), z.insert_sizes())
    assert m["insert_size_metrics"].adapters_read1() == z.adapters_read1()
    np.testing.assert_array_equal(u64(m["dedup_estimator"].duplication_counts()), d.duplication_counts())
    # tests/test_integration.py of the reference: out of sync files and mismatching names
    short = tmp_path / "short.fastq"
    short.write_bytes(b"".join(t2.split(b"\n")[i] + b"\n" for i in range(4 * 10)))
    with pytest.raises(RuntimeError, match="out of sync"):
        driver.run(str(p1), str(short))
    with pytest.raises(RuntimeError, match="out of sync"):
        driver.run(str(short), str(p1))
    renamed = tmp_path / "renamed.fastq"
    lines = t2.split(b"\n")
    lines[4 * 7] = b"@some_other_read/2"
    renamed.write_bytes(b"\n".join(lines))
    with pytest.raises(RuntimeError, match="Mismatching names found! .* some_other_read/2"):
        driver.run(str(p1), str(renamed))


def test_bam_file_and_cli(tmp_path):
    from sequali_amd import driver
    bam = np.load(os.path.join(GOLDEN, "bam_cases.npz"))["dorado_nanopore_100reads_bam"].tobytes()
    path = tmp_path / "reads.bam"
    path.write_bytes(gzip.compress(bam[:100_000]) + gzip.compress(bam[100_000:]))   # two gzip members, like BGZF
    m = driver.run(str(path))
    assert m["sequencing_technology"] == "nanopore"
    assert m["nanostats"].number_of_reads == 100 and m["nanostats"].skipped_reason is None
    assert len(m["adapters"]) == 14
    out = tmp_path / "out.json"
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    subprocess.run([sys.executable, "-m", "sequali_amd", str(path), "--json", str(out), "--raw"], check=True, env=env)
    report = json.loads(out.read_text())
    assert report["metrics"]["number_of_reads"] == 100
    assert report["nanostats"]["number_of_reads"] == 100
    assert report["adapter_counter"]["number_of_sequences"] == 100


# ---- the JSON report (sequali_amd.report): the reference's integration expectations -------------
def _report_of(tmp_path, name1, name2=None, **kw):
    from sequali_amd import driver, report
    from tests.helpers import golden_text
    paths = []
    for k, name in enumerate([name1] + ([name2] if name2 else [])):
        g = golden(name[0])
        p = tmp_path / f"reads{k}.fastq"
        p.write_bytes(golden_text(g, name[1]))
        paths.append(str(p))
    modules = driver.run(*paths, **kw)
    return json.loads(json.dumps(report.report(modules, *paths)))   # must be JSON serialisable


def test_report_simple_fastq(tmp_path):
    """tests/test_integration.py:29-42"""
    r = _report_of(tmp_path, ("ref_simple", "fastq"))
    assert r["summary"]["maximum_length"] == 8
    assert r["summary"]["minimum_length"] == 7
    assert r["summary"]["total_gc_bases"] == 4
    assert r["summary"]["total_bases"] == 22
    assert r["sequence_length_distribution"]["n50"] == 7
    assert r["sequence_length_distribution"]["n90"] == 7
    assert set(r) >= {"meta", "summary", "sequence_length_distribution", "per_sequence_quality_scores",
                      "per_position_base_content", "per_position_n_content", "per_sequence_gc_content",
                      "overrepresented_sequences", "duplication_fractions"}


def test_report_empty_file(tmp_path):
    """tests/test_integration.py:45-57 (there on summary of an empty file)"""
    r = _report_of(tmp_path, ("ref_empty", "fastq"))
    for key in ("maximum_length", "minimum_length", "total_gc_bases", "total_bases"):
        assert r["summary"][key] == 0


def test_report_adapters_only(tmp_path):
    """tests/test_integration.py:97-124 (without the identification of the sequences against the
    contaminant database, which is out of scope)"""
    r = _report_of(tmp_path, ("ref_100_illumina_adapters", "fastq"), overrepresentation_sample_every=1)
    assert r["summary"]["maximum_length"] == 33
    assert r["summary"]["minimum_length"] == 33
    assert r["summary"]["total_gc_bases"] == 1700
    assert r["summary"]["total_bases"] == 3300
    for adapter_name, quantities in r["adapter_content"]["adapter_content"]:
        if adapter_name == "Illumina Universal Adapter":
            assert quantities == [100.0] * 33
        assert len(quantities) == 33
    o = r["overrepresented_sequences"]
    assert o["total_sequences"] == 100 and o["sampled_sequences"] == 100 and o["total_fragments"] == 200
    assert o["overrepresented_sequences"] and all(d["count"] == 100 for d in o["overrepresented_sequences"])


def test_report_paired_end(tmp_path):
    """tests/test_integration.py:203-211"""
    r = _report_of(tmp_path, ("ref_LTB_paired", "fastq1"), ("ref_LTB_paired", "fastq2"))
    assert "summary_read2" in r and "insert_size_metrics" in r and "overrepresented_sequences_read2" in r
    assert r["summary"]["total_reads"] == r["summary_read2"]["total_reads"] == 1000
    assert r["summary"]["read_pair_info"] == "Read 1" and r["summary_read2"]["read_pair_info"] == "Read 2"
    assert sum(r["insert_size_metrics"]["insert_sizes"]) == 1000


REFERENCE_KEYS = {   # report_modules.py:2412-2427 (NAME_TO_CLASS) + the _read2 variants a paired run adds
    "meta", "summary", "per_position_mean_quality_and_spread", "per_position_quality_distribution",
    "sequence_length_distribution", "per_position_base_content", "per_position_n_content",
    "per_sequence_gc_content", "per_sequence_quality_scores", "adapter_content", "per_tile_quality",
    "duplication_fractions", "overrepresented_sequences", "nanopore_metrics"}


def test_report_has_the_reference_key_set_and_the_quality_modules_agree_with_numpy(tmp_path):
    """all sixteen keys of the reference's JSON (report_modules.py:2412-2427) on a paired run, and the
    modules added in round 3 recomputed here with numpy from the same getters: the quality
    distribution is a column-normalised table, the 'mean' series is -10 log10 of the count-weighted
    bin error rates (:64-67, :776-779), bottom-50 % <= mean <= top-50 %; per-tile rows are phreds
    minus the column mean over tiles (:1507-1533); the overlap adapters are the most frequent one of
    every length (:2295-2310)"""
    r = _report_of(tmp_path, ("ref_LTB_paired", "fastq1"), ("ref_LTB_paired", "fastq2"))
    # a paired run has no AdapterCounter (__main__.py:222-242: the overlap of the mates finds the adapters)
    assert set(r) >= (REFERENCE_KEYS - {"adapter_content"}) | {"insert_size_metrics", "adapter_content_from_overlap"}
    for suffix in ("", "_read2"):
        for key in ("summary", "per_position_mean_quality_and_spread", "per_position_quality_distribution",
                    "per_tile_quality", "sequence_length_distribution", "per_position_base_content"):
            assert key + suffix in r, key + suffix
    from sequali_amd import driver
    from tests.helpers import golden_text
    g = golden("ref_LTB_paired")
    p1, p2 = tmp_path / "a.fastq", tmp_path / "b.fastq"
    p1.write_bytes(golden_text(g, "fastq1")); p2.write_bytes(golden_text(g, "fastq2"))
    m = driver.run(str(p1), str(p2))
    qc = m["metrics"]
    phred = np.array(qc.phred_count_table(), dtype=np.float64).reshape(-1, 12)   # 125 positions: one per range
    dist = np.array(r["per_position_quality_distribution"]["series"])            # [12][ranges]
    assert dist.shape == (12, len(phred))
    np.testing.assert_allclose(dist.T, phred / phred.sum(axis=1, keepdims=True), rtol=1e-12)
    np.testing.assert_allclose(dist.sum(axis=0), 1.0, rtol=1e-12)
    err = np.array([sum(10 ** (-q / 10) for q in range(4 * b, 4 * b + 4)) / 4 for b in range(12)])
    series = dict((k, np.array(v)) for k, v in r["per_position_mean_quality_and_spread"]["percentiles"])
    np.testing.assert_allclose(series["mean"], -10 * np.log10((phred * err).sum(axis=1) / phred.sum(axis=1)), rtol=1e-12)
    assert (series["bottom 50%"] <= series["mean"] + 1e-9).all() and (series["mean"] <= series["top 50%"] + 1e-9).all()
    assert (series["bottom 1%"] <= series["bottom 25%"] + 1e-9).all() and (series["top 25%"] <= series["top 1%"] + 1e-9).all()
    assert len(r["per_position_mean_quality_and_spread"]["front_percentiles"][0][1]) == 100
    # per tile: phred of the mean error per tile and position, minus the mean over the tiles
    tiles = m["per_tile_quality"].get_tile_counts()
    ph = np.array([-10 * np.log10(np.array(e) / np.maximum(np.array(c, dtype=np.float64), 1)) for _, e, c in tiles])
    rows = r["per_tile_quality"]["normalized_per_tile_averages"]
    assert [t for t, _ in rows] == [str(t) for t, _, _ in tiles]
    np.testing.assert_allclose(np.array([v for _, v in rows]), ph - ph.mean(axis=0), rtol=1e-9, atol=1e-9)
    assert r["per_tile_quality"]["skipped_reason"] is None
    # adapters from the overlap: one per length, the most frequent one
    isz = m["insert_size_metrics"]
    best = {}
    for a, c in isz.adapters_read1():
        if len(a) not in best or c > best[len(a)][1]:
            best[len(a)] = (a, c)
    got = r["adapter_content_from_overlap"]["adapters_read1"]
    assert [len(a) for a, _ in got] == sorted(best) and all(best[len(a)][1] == c for a, c in got)
    assert r["adapter_content_from_overlap"]["total_reads"] == 1000
    assert r["nanopore_metrics"]["skipped_reason"] is not None and r["nanopore_metrics"]["time_reads"] == []


def test_report_nanopore_metrics(tmp_path):
    """NanoStatsReport.from_nanostats (report_modules.py:1951-2041) on the reference's nanopore reads:
    every read lands in one time slot and one translocation-speed bin, the per-channel bases add
    up to the bases of the file"""
    r = _report_of(tmp_path, ("ref_100_nanopore", "fastq"))
    n = r["nanopore_metrics"]
    assert n["skipped_reason"] is None and n["total_reads"] == 100
    assert sum(n["time_reads"]) == 100 and sum(n["translocation_speed"]) <= 100
    assert sum(n["time_bases"]) == sum(n["per_channel_bases"].values()) == r["summary"]["total_bases"]
    assert len(n["x_labels"]) == len(n["time_reads"]) == len(n["time_active_channels"])
    assert all(len(s) == len(n["time_reads"]) for s in n["qual_percentages_over_time"])
    assert r["per_tile_quality"]["skipped_reason"] is not None
