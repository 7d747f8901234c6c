"""Which kernel takes which batch: the dispatchers choose per batch and fall back to another kernel
without a word when a build does not fit (register spills, LDS); the numbers in BASELINE / DESIGN are those of the
routes asserted here (sq_last_route).  Shapes: the bench's configurations at a small size."""
import pytest

pytestmark = pytest.mark.gpu


def route_of(run):
    from sequali_amd._lib import context, lib
    lib().sq_route_reset(context())
    run()
    return (lib().sq_last_route(context()) or b"").decode()


def test_headline_route_is_a_wave_per_stream():
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, synth
    dev = synth.device_array(synth.ILLUMINA, 0, 200_000)
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    r = route_of(lambda: (f.add_record_array(dev), f.qc_metrics.flush()))
    assert r.split("+")[0] == "k_span<5,AD,uniform,split>", r


def test_qcmetrics_alone_is_one_wave_for_both_streams():
    from sequali_amd import QCMetrics, synth
    dev = synth.device_array(synth.ILLUMINA, 0, 200_000)
    q = QCMetrics()
    r = route_of(lambda: (q.add_record_array(dev), q.flush()))
    assert r.split("+")[0] == "k_span<5,QC,uniform,both>", r


@pytest.mark.parametrize("L,with_adapters,alone", [(224, "k_span<7,AD,uniform,split>", "k_span<7,QC,uniform,split>"),
                                                   (250, "k_span<8,AD,uniform,split>", "k_span<8,QC,uniform,split>")])
def test_longer_illumina_reads(L, with_adapters, alone):
    """reads of up to 256 bases take k_span with and without adapters, a wave per stream in both cases from 161 bases on (QCMetrics
    alone: 13-14 % ahead of one wave for both streams there, profiles/r5/exp_split_qc.txt; 225-256 with adapters went to k_wide until round 5:
    at 12 waves per CU instead of 8 the 8-window build is 10-14 % ahead, scripts/exp_len2.sh)"""
    from sequali_amd import AdapterCounter, FastqRecordArrayView, FusedPass, QCMetrics, synth
    import numpy as np
    from tests.helpers import split_fastq
    rng = np.random.default_rng(3)
    n = 4096
    seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L)).view(f"S{L}").ravel()
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seqs[i], b"I" * L) for i in range(n))
    buf, metas = split_fastq(text)
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    arr = FastqRecordArrayView._from_buffer(buf, metas)
    r = route_of(lambda: (f.add_record_array(arr), f.qc_metrics.flush()))
    assert r.split("+")[0] == with_adapters, r
    q = QCMetrics()
    r = route_of(lambda: (q.add_record_array(arr), q.flush()))
    assert r.split("+")[0] == alone, r


def test_config3_route():
    """PerTileQuality rides in QCMetrics' pass (sq_pair.hip): reads that come tile by tile leave the table to the pass
    too, reads of random tiles only their tile ids (the table: k_ptspan)"""
    from sequali_amd import FusedPass, InsertSizeMetrics, PerTileQuality, QCMetrics, synth
    d1 = synth.device_array(synth.ILLUMINA_BY_TILE, 0, 100_000)
    d2 = synth.device_array(synth.ILLUMINA_R2_BY_TILE, 0, 100_000)
    from tests.helpers import with_env
    fa, z = FusedPass(QCMetrics(), None, PerTileQuality()), InsertSizeMetrics()
    r = route_of(lambda: with_env({"SQ_PT_FUSED": "0"}, lambda: (fa.add_record_array(d1), fa.qc_metrics.flush())))
    assert r == "k_span<5,QC,uniform,both>+k_ptspan<5>", r          # the passes of round 2
    fa = FusedPass(QCMetrics(), None, PerTileQuality())
    r = route_of(lambda: (fa.add_record_array(d1), fa.qc_metrics.flush()))
    assert r == "k_span<5,QCPT,uniform,both>+k_pt_fold", r          # the default since round 5
    fr = FusedPass(QCMetrics(), None, PerTileQuality())
    dr = synth.device_array(synth.ILLUMINA, 0, 100_000)
    r = route_of(lambda: (fr.add_record_array(dr), fr.qc_metrics.flush()))
    assert r == "k_span<5,QCPT,uniform,both>+k_ptspan<5>", r
    r = route_of(lambda: z.add_record_array_pair(d1, d2))
    assert r.startswith("k_isz_span<5>"), r


def test_ragged_route():
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, synth
    dev = synth.device_array(synth.ILLUMINA, 0, 2_000_000)
    _lib.check(_lib.lib().sq_synth_trim(dev._batch.handle, 7, 50))
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    r = route_of(lambda: (f.add_record_array(dev), f.qc_metrics.flush()))
    parts = r.split("+")
    assert parts[0] == "k_span_scatter", r
    assert sorted(parts[1:]) == [f"k_span<{nw},AD,sorted,both>" for nw in (2, 3, 4, 5)], r


def test_long_read_route():
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, synth
    dev = synth.device_array(synth.NANOPORE, 0, 6000)
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.NANOPORE_PROBES)))
    r = route_of(lambda: (f.add_record_array(dev), f.qc_metrics.flush()))
    assert r == "k_read_sums<qualities>+k_span<8,AD,long>", r
