"""The reference's own call pattern: FastqParser at its default 128 KiB (about 380 reads per
array, _qcmodule.c:915) and every module called once per array (__main__.py:279-306).  The shim
stages such arrays and launches once per 64 MiB and module; the results are those of the oracle
on the whole file."""
import io
import warnings

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import golden, golden_text, split_fastq

pytestmark = pytest.mark.gpu


def u64(a):
    return np.array(a, dtype=np.uint64)


def test_default_buffer_single_end_all_modules_few_launches():
    from sequali_amd import (AdapterCounter, DedupEstimator, FastqParser, NanoStats, OverrepresentedSequences,
                             PerTileQuality, QCMetrics, _qc, synth)
    n = 200_000
    text = synth.illumina_fastq(0, n)
    probes = list(synth.ILLUMINA_PROBES)
    qc, ad, pt, ov, dd, ns = (QCMetrics(), AdapterCounter(probes), PerTileQuality(), OverrepresentedSequences(),
                              DedupEstimator(), NanoStats())
    before = dict(_qc.staging_stats)
    arrays, records = 0, 0
    kept = []
    for arr in FastqParser(io.BytesIO(text)):          # default initial_buffersize
        arrays += 1
        for mod in (qc, ad, pt, ov, dd, ns):
            mod.add_record_array(arr)
        if arrays in (1, 100):
            kept.append((records, arr))
        records += len(arr)
    assert arrays > 400                                  # ~128 KiB of text each
    # accumulated_error_rate of an array from the middle of a staging block (:2126)
    buf, metas = split_fastq(text)
    rq, ra, rp, ro, rd = (oracle.QCMetrics(), oracle.AdapterCounter(probes), oracle.PerTileQuality(),
                          oracle.OverrepresentedSequences(), oracle.DedupEstimator())
    for r in (rq, ra, rp, ro, rd):
        r.add(buf, metas)
    for first, arr in kept:
        got = arr.accumulated_error_rates()
        np.testing.assert_array_equal(got.view(np.uint64), metas["accumulated_error_rate"][first:first + len(arr)].view(np.uint64))
    assert qc.number_of_reads == rq.number_of_reads == n and qc.max_length == rq.max_length
    np.testing.assert_array_equal(u64(qc.base_count_table()), rq.base_count_table())
    np.testing.assert_array_equal(u64(qc.phred_count_table()), rq.phred_count_table())
    np.testing.assert_array_equal(u64(qc.end_anchored_base_count_table()), rq.end_anchored_base_count_table())
    np.testing.assert_array_equal(u64(qc.end_anchored_phred_count_table()), rq.end_anchored_phred_count_table())
    np.testing.assert_array_equal(u64(qc.gc_content()), rq.gc_content())
    np.testing.assert_array_equal(u64(qc.phred_scores()), rq.phred_scores())
    for (_, f, r), (_, fr, rr) in zip(ad.get_counts(), ra.get_counts()):
        np.testing.assert_array_equal(u64(f), fr)
        np.testing.assert_array_equal(u64(r), rr)
    assert pt.number_of_reads == rp.number_of_reads
    for (t, e, c), (tr, er, cr) in zip(pt.get_tile_counts(), rp.get_tile_counts()):
        assert t == tr
        np.testing.assert_array_equal(u64(c), cr)
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6, atol=0)
    assert ov.sequence_counts() == ro.sequence_counts() and ov.total_fragments == ro.total_fragments
    np.testing.assert_array_equal(u64(dd.duplication_counts()), rd.duplication_counts())
    assert ns.skipped_reason is not None and ns.number_of_reads == 0      # no nanopore header: off at the first read
    # 70 MB of text: an 8 MiB block (a parser starts small) and one or two of 64 MiB, at most one launch per block
    # and module (NanoStats stops at its first)
    blocks = _qc.staging_stats["blocks"] - before["blocks"]
    runs = _qc.staging_stats["runs"] - before["runs"]
    assert 2 <= blocks <= 3 and runs <= 6 * blocks, (blocks, runs)


def test_default_buffer_paired_reference_files():
    """the reference's 1000-pair test files through its driver's paired loop at the default
    buffer: QCMetrics x 2, PerTileQuality x 2, paired DedupEstimator and InsertSizeMetrics against
    the goldens captured from the compiled reference"""
    from sequali_amd import DedupEstimator, FastqParser, InsertSizeMetrics, PerTileQuality, QCMetrics
    g = golden("ref_LTB_paired")
    t1, t2 = golden_text(g, "fastq1"), golden_text(g, "fastq2")
    q1, q2, p1, p2, dd, isz = QCMetrics(), QCMetrics(), PerTileQuality(), PerTileQuality(), DedupEstimator(
        front_sequence_offset=0, back_sequence_offset=0), InsertSizeMetrics()
    r1, r2 = FastqParser(io.BytesIO(t1)), FastqParser(io.BytesIO(t2))
    for a1 in r1:
        a2 = r2.read(len(a1))
        assert a1.is_mate(a2)
        q1.add_record_array(a1); p1.add_record_array(a1)
        q2.add_record_array(a2); p2.add_record_array(a2)
        dd.add_record_array_pair(a1, a2)
        isz.add_record_array_pair(a1, a2)
    for q, pre in ((q1, "qc1_"), (q2, "qc2_")):
        assert q.number_of_reads == int(g[pre + "number_of_reads"])
        np.testing.assert_array_equal(u64(q.base_count_table()), g[pre + "base"])
        np.testing.assert_array_equal(u64(q.phred_count_table()), g[pre + "phred"])
        np.testing.assert_array_equal(u64(q.gc_content()), g[pre + "gc"])
        np.testing.assert_array_equal(u64(q.phred_scores()), g[pre + "phred_scores"])
    np.testing.assert_array_equal(u64(isz.insert_sizes()), g["is_insert_sizes"])
    np.testing.assert_array_equal(u64(dd.duplication_counts()), g["dd_counts_slot_order"])
    assert dd._modulo_bits == int(g["dd_modulo_bits"]) and isz.total_reads == int(g["is_total_reads"])


def test_getter_in_the_middle_of_a_block_seals_it_and_the_parser_goes_on():
    """a getter between two arrays needs the state of everything handed over so far: the parser's
    open staging block is sealed behind its last array (sq_feeder_seal), uploaded and counted, and
    the parser continues in a new block; arrays the caller still holds keep answering (their
    bytes come from the pinned block while it is there, from HBM afterwards)"""
    from sequali_amd import AdapterCounter, FastqParser, FusedPass, QCMetrics, synth
    n = 60_000
    text = synth.illumina_fastq(0, n)
    buf, metas = split_fastq(text)
    probes = list(synth.ILLUMINA_PROBES)
    qc, ad = QCMetrics(), AdapterCounter(probes)
    f = FusedPass(qc, ad)
    seen, kept = 0, []
    for k, arr in enumerate(FastqParser(io.BytesIO(text))):
        f.add_record_array(arr)
        if k % 23 == 4:
            kept.append((seen, arr))
        seen += len(arr)
        if k % 17 == 5:
            assert qc.number_of_reads == seen and ad.number_of_sequences == seen     # seals the open block
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(probes)
    rq.add(buf, metas)
    ra.add(buf, metas)
    assert seen == n and qc.number_of_reads == n
    np.testing.assert_array_equal(u64(qc.base_count_table()), rq.base_count_table())
    np.testing.assert_array_equal(u64(qc.phred_count_table()), rq.phred_count_table())
    np.testing.assert_array_equal(u64(qc.phred_scores()), rq.phred_scores())
    for (_, fw, rv), (_, fr, rr) in zip(ad.get_counts(), ra.get_counts()):
        np.testing.assert_array_equal(u64(fw), fr)
        np.testing.assert_array_equal(u64(rv), rr)
    for first, arr in kept:       # long after their blocks were sealed, some after the pinned copy was recycled
        m = metas[first:first + len(arr)]
        np.testing.assert_array_equal(arr.accumulated_error_rates().view(np.uint64), m["accumulated_error_rate"].view(np.uint64))
        i = len(arr) // 2
        want = buf[int(m[i]["record_start"]) + int(m[i]["sequence_offset"]):][:int(m[i]["sequence_length"])]
        assert arr[i].sequence().encode() == want


@pytest.mark.gpu
def test_sealed_blocks_return_to_the_pinned_pool_without_the_cycle_collector():
    """a block that has been sealed and uploaded (a module counted its arrays) must not be kept alive by its own
    device array: with the cycle collector off, the second and third parser over the same text lock no new
    memory (sq_feeder_debug_times counts the pool's fresh allocations)"""
    import ctypes as C
    import gc
    import io
    from sequali_amd import FastqParser, QCMetrics
    from sequali_amd._lib import lib
    text = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(50000))
    out = (C.c_double * 4)()
    gc.collect()
    gc.disable()
    try:
        for rep in range(3):
            lib().sq_feeder_debug_times(out, 1)
            q = QCMetrics()
            for a in FastqParser(io.BytesIO(text), 8192):
                q.add_record_array(a)
            assert q.number_of_reads == 50000
            del q, a
            lib().sq_feeder_debug_times(out, 0)
            if rep:
                assert out[3] == 0, "a later parser had to lock staging memory afresh"
    finally:
        gc.enable()


def test_an_array_staged_by_one_module_and_run_on_its_own_by_another_has_one_copy_in_hbm():
    """QCMetrics_add_meta leaves accumulated_error_rate IN the array's metas (_qcmodule.c:2126) and NanoStats reads it there
    (:5314).  A module that defers its work copies a small array into a staging block; a pair whose MATE is too large to
    stage sends the pair's module -- and with it the small array -- down the unstaged path.  The array's records in HBM must
    then be the block's: with a second upload the staged QCMetrics pass wrote the error rates into one copy and NanoStats
    read the other (all 0.0; found by scripts/fuzz.py 200 55, round 5)."""
    from sequali_amd import FastqRecordArrayView, FusedPass, InsertSizeMetrics, NanoStats, QCMetrics
    rng = np.random.default_rng(55)

    def batch(n, U):
        names = [f"read{i} ch={i % 512} start_time=2021-09-30T11:34:{i % 60:02d}Z" for i in range(n)]
        seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=U).tobytes().decode() for _ in range(n)]
        quals = [(rng.integers(0, 94, size=U) + 33).astype(np.uint8).tobytes().decode() for _ in range(n)]
        return oracle.make_batch(names, seqs, quals)

    n = 3000
    b1, m1 = batch(n, 200)
    b2, m2 = batch(n, 1500)
    assert len(b1) < (8 << 20) < len(b2)          # the mate alone is beyond what is staged
    rq, rz, rn = oracle.QCMetrics(), oracle.InsertSizeMetrics(), oracle.NanoStats()
    q, z, ns = QCMetrics(), InsertSizeMetrics(), NanoStats()
    f = FusedPass(q, None, None)
    for lo, hi in ((0, 1700), (1700, n)):
        x1, x2 = m1[lo:hi].copy(), m2[lo:hi].copy()
        rq.add(b1, x1); rz.add_pair(b1, x1, b2, x2); rn.add(b1, x1)
        a1 = FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy())
        a2 = FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy())
        f.add_record_array(a1)               # staged
        z.add_record_array_pair(a1, a2)      # not staged: a2 is too large
        ns.add_record_array(a1)              # a1 is in HBM by now: not staged either
        np.testing.assert_array_equal(a1.accumulated_error_rates().view(np.uint64), x1["accumulated_error_rate"].view(np.uint64))
    gi, ri = ns.nano_infos(), rn.nano_infos()
    np.testing.assert_array_equal(gi["cumulative_error_rate"].view(np.uint64), ri["cumulative_error_rate"].view(np.uint64))
    np.testing.assert_array_equal(np.array(q.phred_scores(), np.uint64), rq.phred_scores())
    np.testing.assert_array_equal(np.array(z.insert_sizes(), np.uint64), rz.insert_sizes())


def test_batches_of_one_block_run_in_call_order():
    """Arrays of several streams staged into ONE block leave a pass several stretches of it to run.  They must run in call
    order: InsertSizeMetrics keeps the first max_adapters remainders in pair order (_qcmodule.c:5570-5611) and
    PerTileQuality stops for good at the first header that does not parse (:3137-3148).  (The drain was re-entrant through
    the QCMetrics objects a pass feeds and ran the stretches last first: scripts/fuzz.py 200 11, iteration 192, round 5.)"""
    from sequali_amd import DedupEstimator, FastqRecordArrayView, FusedPass, InsertSizeMetrics, PairedPass, PerTileQuality, QCMetrics
    from tests.test_gpu_pair import _pair_batches, _runs
    rng = np.random.default_rng(192)
    n = 1500
    tiles = _runs(rng, n, [40, 300], [1101, 1102, 2203])
    (b1, m1), (b2, m2) = _pair_batches(rng, n, 100, 90, tiles)
    # a header that does not parse in the SECOND of three stretches
    bad = 700
    names_start = int(m1["record_start"][bad])
    b1 = bytearray(b1)
    b1[names_start:names_start + 4] = b"xxxx"
    for k in range(names_start, names_start + int(m1["name_length"][bad])):
        if b1[k] == ord(":"):
            b1[k] = ord("_")
    b1 = bytes(b1)
    cuts = [0, 500, 1000, n]
    ref = (oracle.QCMetrics(), oracle.PerTileQuality(), oracle.QCMetrics(), oracle.PerTileQuality(), oracle.InsertSizeMetrics(2))
    got = (QCMetrics(), PerTileQuality(), QCMetrics(), PerTileQuality(), InsertSizeMetrics(2))
    pp = PairedPass(*got)
    fq, fp = QCMetrics(), PerTileQuality()
    ff = FusedPass(fq, None, fp)
    other = DedupEstimator()
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        x1, x2 = m1[lo:hi].copy(), m2[lo:hi].copy()
        ref[0].add(b1, x1); ref[1].add(b1, x1); ref[2].add(b2, x2); ref[3].add(b2, x2); ref[4].add_pair(b1, x1, b2, x2)
        pp.add_record_array_pair(FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy()), FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy()))
        ff.add_record_array(FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy()))
        other.add_record_array(FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy()))    # another stream in between
    assert got[4].adapters_read1() == ref[4].adapters_read1() and got[4].adapters_read2() == ref[4].adapters_read2()
    np.testing.assert_array_equal(np.array(got[4].insert_sizes(), np.uint64), ref[4].insert_sizes())
    assert ref[1].number_of_reads == bad        # the reference counted the reads in front of the bad header and nothing behind
    for p in (got[1], fp):
        assert p.number_of_reads == ref[1].number_of_reads and p.skipped_reason is not None
        for (t, e, c), (tr, er, cr) in zip(p.get_tile_counts(), ref[1].get_tile_counts()):
            assert t == tr
            np.testing.assert_array_equal(np.array(c, np.uint64), cr)
            np.testing.assert_allclose(np.array(e), er, rtol=1e-6)
    assert got[3].number_of_reads == ref[3].number_of_reads == n
    np.testing.assert_array_equal(np.array(fq.phred_scores(), np.uint64), ref[0].phred_scores())


def _pool_counts():
    import ctypes as C
    from sequali_amd import _lib
    out = (C.c_uint64 * 4)()
    _lib.lib().sq_pool_counts(_lib.context(), out)
    return list(out)


def test_a_second_pass_over_a_file_allocates_no_device_memory_and_qcmetrics_holds_few_blocks(monkeypatch):
    """Round 6: every staging block of a file used to be a fresh hipMalloc -- the context's pool was full of sizes nobody asked
    for, and QCMetrics kept every block's copy in HBM until its flush (an invalid phred character is only found there).  Now
    the pool evicts what has lain longest, freed arrays hand their blocks back behind an event, and QCMetrics finds out
    without waiting which passes have ended clean (sq_qcmetrics_poll).  Small blocks so that a small file has many."""
    from sequali_amd import AdapterCounter, FastqParser, FusedPass, QCMetrics, _qc, synth
    monkeypatch.setattr(_qc, "_STAGE_LIMIT", 1 << 20)
    n = 60_000
    text = synth.illumina_fastq(0, n)
    buf, metas = split_fastq(text)
    want = oracle.QCMetrics()
    want.add(buf, metas)
    for rep in range(3):
        before = _pool_counts()
        blocks = _qc.staging_stats["blocks"]
        q = QCMetrics()
        f = FusedPass(q, AdapterCounter(list(synth.ILLUMINA_PROBES)), None)
        held = 0
        for a in FastqParser(io.BytesIO(text)):
            f.add_record_array(a)
            held = max(held, len(q._pending))
        np.testing.assert_array_equal(u64(q.base_count_table()), want.base_count_table())
        np.testing.assert_array_equal(u64(q.phred_count_table()), want.phred_count_table())
        assert _qc.staging_stats["blocks"] - blocks >= 15
        assert held <= 6, f"QCMetrics held {held} staging blocks' arrays"
        del q, f, a
        after = _pool_counts()
        if rep:
            assert after[0] == before[0], f"pass {rep}: {after[0] - before[0]} hipMalloc calls"
            assert after[1] - before[1] <= 4, f"pass {rep}: {after[1] - before[1]} hipFree calls"    # (stale sizes leave the pool)


def test_an_invalid_phred_character_behind_blocks_that_were_let_go(monkeypatch):
    """the arrays in front of the offending block have been let go of by then (sq_qcmetrics_poll); the flush still raises, and
    the tables are the reference's behind the call that raised (:2102-2105)"""
    from sequali_amd import FastqParser, QCMetrics, _qc, synth
    monkeypatch.setattr(_qc, "_STAGE_LIMIT", 1 << 20)
    n = 40_000
    text = bytearray(synth.illumina_fastq(0, n))
    buf, metas = split_fastq(bytes(text))
    bad = 33_333
    at = int(metas["record_start"][bad]) + int(metas["qualities_offset"][bad]) + 7
    text[at] = 0x1F            # not a phred character
    buf, metas = split_fastq(bytes(text))
    q = QCMetrics()
    arrays = []
    with pytest.raises(ValueError, match="Not a valid phred character"):
        for a in FastqParser(io.BytesIO(bytes(text))):
            arrays.append(len(a))
            q.add_record_array(a)
        q.flush()
    # the reference raises inside the offender's call: that array counted up to the read in front of it, the offender's bases
    # and the phreds in front of the character (:2102-2105).  Here the error is found at the flush, so the caller's loop has
    # handed in the arrays behind it too: a caller of the reference who caught the exception and went on
    want = oracle.QCMetrics()
    first, raised = 0, 0
    for k in arrays:
        try:
            want.add(buf, metas[first:first + k])
        except ValueError:
            raised += 1
        first += k
    assert raised == 1
    assert q.number_of_reads == want.number_of_reads
    np.testing.assert_array_equal(u64(q.base_count_table()), want.base_count_table())
    np.testing.assert_array_equal(u64(q.phred_count_table()), want.phred_count_table())


@pytest.mark.parametrize("early,walker", [("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")])
def test_blocks_sent_while_they_fill_and_when_they_are_sealed(monkeypatch, early, walker):
    """the four ways a staging block reaches HBM (by the walker as the text arrives, by the workers piece by piece, with one
    copy when it is sealed; records split by the walker or window by window): the same tables"""
    from sequali_amd import AdapterCounter, FastqParser, FusedPass, PerTileQuality, QCMetrics, _qc, synth
    monkeypatch.setenv("SQ_FEED_EARLY", early)
    monkeypatch.setenv("SQ_FEED_WALKER", walker)
    monkeypatch.setattr(_qc, "_STAGE_LIMIT", 3 << 20)
    n = 50_000
    text = synth.illumina_fastq(0, n)
    buf, metas = split_fastq(text)
    rq, rp = oracle.QCMetrics(), oracle.PerTileQuality()
    rq.add(buf, metas)
    rp.add(buf, metas)
    q, p = QCMetrics(), PerTileQuality()
    f = FusedPass(q, AdapterCounter(list(synth.ILLUMINA_PROBES)), p)
    for i, a in enumerate(FastqParser(io.BytesIO(text))):
        f.add_record_array(a)
        if i == 40:
            assert q.number_of_reads > 0     # a getter in the middle of a block: sealed with its tail still on its way
    np.testing.assert_array_equal(u64(q.base_count_table()), rq.base_count_table())
    np.testing.assert_array_equal(u64(q.phred_count_table()), rq.phred_count_table())
    np.testing.assert_array_equal(u64(q.gc_content()), rq.gc_content())
    assert p.number_of_reads == rp.number_of_reads == n
    for (t, e, c), (tr, er, cr) in zip(p.get_tile_counts(), rp.get_tile_counts()):
        assert t == tr
        np.testing.assert_array_equal(u64(c), cr)
        np.testing.assert_allclose(np.array(e), er, rtol=1e-6, atol=0)
