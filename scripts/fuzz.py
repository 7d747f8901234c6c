#!/usr/bin/env python3
"""Randomised differential run: HIP modules against the oracle on random batches, random
module parameters and random batch splits.  python scripts/fuzz.py [iterations] [seed]"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from sequali_amd import (AdapterCounter, DedupEstimator, FastqRecordArrayView, FusedPass, InsertSizeMetrics,  # noqa: E402
                         NanoStats, OverrepresentedSequences, PairedPass, PerTileQuality, QCMetrics)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # run this iteration alone (a failure seen before)
SKIP = set(filter(None, os.environ.get("FUZZ_SKIP", "").split(",")))   # modules left out of the GPU side's calls AND of the checks (bisecting a failure): o,d,dp,z,n


def u64(a):
    return np.array(a, dtype=np.uint64)


def make(rng, n, max_len, uniform, illumina, adapters):
    names, seqs, quals = [], [], []
    U = int(rng.integers(1, max_len + 1))
    pool = []
    for i in range(n):
        L = U if uniform else int(rng.integers(0, max_len + 1))
        if pool and rng.random() < 0.25:
            s = pool[int(rng.integers(0, len(pool)))]
            s = (s * (L // max(len(s), 1) + 1))[:L]
        else:
            s = rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), size=L,
                           p=[.22, .22, .22, .22, .02, .025, .025, .025, .025]).tobytes().decode()
            if len(pool) < 50:
                pool.append(s)
        if adapters and rng.random() < 0.3:
            w = adapters[int(rng.integers(0, len(adapters)))]
            if len(w) <= L:
                at = int(rng.integers(0, L - len(w) + 1))
                s = s[:at] + w + s[at + len(w):]
        L = len(s)      # (an empty read in the pool repeats to an empty read: the qualities must follow, or the text is no FASTQ)
        q = (rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes().decode()
        tile = int(rng.choice([1101, 1102, 2205, 7, 99239]))
        names.append(f"M:1:F:{i % 4}:{tile}:{i}:{L} 1:N:0:X" if illumina else f"read{i} ch={i % 512} start_time=2021-09-30T11:34:{i % 60:02d}Z")
        seqs.append(s)
        quals.append(q)
    return oracle.make_batch(names, seqs, quals)


if os.environ.get("FUZZ_TRACE"):   # which pass runs when, on which metas in HBM
    import sequali_amd._qc as _Q
    for _name in ("FusedPass", "NanoStats", "InsertSizeMetrics", "QCMetrics", "DedupEstimator", "OverrepresentedSequences", "AdapterCounter", "PerTileQuality"):
        _cls = getattr(_Q, _name)
        def _wrap(orig, name, pair):
            def run(self, arr, *a):
                print(f"   RUN{'_PAIR' if pair else ''} {name} on {len(arr)} records, device metas at {_Q.lib().sq_batch_device_metas(arr._device().handle):#x}", flush=True)
                return orig(self, arr, *a)
            return run
        if "_run" in _cls.__dict__: _cls._run = _wrap(_cls._run, _name, False)
        if "_run_pair" in _cls.__dict__: _cls._run_pair = _wrap(_cls._run_pair, _name, True)

ADS = [["AGATCGGAAGAG", "CTGTCTCTTATA", "GGGGGGGGGGGG"], ["ACG", "NN", "GTAC", "TTTTTTTT"], ["ACGT" * 16, "A" * 40],
       ["ACGGTCATTGCACTTAGGCATCGAT", "TGACCGTTAGCAGGATCCTA", "GTTACCAGTCAGGA"]]   # (the last: 14-25 characters, the six-dword builds of k_span)
def check(got, pair, ref, tag):
    """every getter of the GPU objects against the oracle's; `tag` names the feeding path in a failure"""
    try:
        _check(got, pair, ref)
    except AssertionError as e:
        raise AssertionError((tag,) + tuple(e.args)) from None


def _check(got, pair, ref):
    g, r = got["q"], ref["q"]
    assert (g.number_of_reads, g.max_length) == (r.number_of_reads, r.max_length), ("qc counters", g.number_of_reads, g.max_length, r.number_of_reads, r.max_length)
    for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                 "end_anchored_phred_count_table", "gc_content", "phred_scores"):
        assert np.array_equal(u64(getattr(g, name)()), getattr(r, name)()), name
    for (_, f, rv), (_, fr, rr) in zip(got["a"].get_counts(), ref["a"].get_counts()):
        assert np.array_equal(u64(f), fr) and np.array_equal(u64(rv), rr), "adapter"
    assert got["p"].number_of_reads == ref["p"].number_of_reads, ("pertile reads", got["p"].number_of_reads, ref["p"].number_of_reads)
    for (t, e, c), (tr, er, cr) in zip(got["p"].get_tile_counts(), ref["p"].get_tile_counts()):
        assert t == tr and np.array_equal(u64(c), cr) and np.allclose(np.array(e), er, rtol=1e-6), "pertile"
    if "o" not in SKIP:
        assert got["o"].sequence_counts() == ref["o"].sequence_counts(), "overrep"
        assert got["o"].total_fragments == ref["o"].total_fragments, "overrep total_fragments"
    for k in [k for k in ("d", "dp") if k not in SKIP]:
        assert got[k]._modulo_bits == ref[k]._modulo_bits, (k, "modulo bits", got[k]._modulo_bits, ref[k]._modulo_bits)
        assert np.array_equal(u64(got[k].duplication_counts()), ref[k].duplication_counts()), k
    if "z" in SKIP: got["z"] = ref["z"]
    if "n" in SKIP: got["n"] = ref["n"]
    gz, rz = u64(got["z"].insert_sizes()), ref["z"].insert_sizes()
    assert np.array_equal(gz, rz), ("insert sizes", len(gz), len(rz), [(int(i), int(gz[i]) if i < len(gz) else None, int(rz[i]) if i < len(rz) else None)
                                                                        for i in range(max(len(gz), len(rz))) if i >= len(gz) or i >= len(rz) or gz[i] != rz[i]][:8])
    assert got["z"].adapters_read1() == ref["z"].adapters_read1(), "insert size adapters of read 1"
    assert got["z"].adapters_read2() == ref["z"].adapters_read2(), "insert size adapters of read 2"
    assert got["n"].number_of_reads == ref["n"].number_of_reads, "nanostats reads"
    gi, ri = got["n"].nano_infos(), ref["n"].nano_infos()
    badn = np.nonzero(gi["cumulative_error_rate"].view(np.uint64) != ri["cumulative_error_rate"].view(np.uint64))[0]
    assert len(badn) == 0, ("nanostats error rates", len(badn), badn[:6].tolist(), badn[-3:].tolist(), gi["cumulative_error_rate"][badn[:3]].tolist(), ri["cumulative_error_rate"][badn[:3]].tolist())
    assert np.array_equal(gi["start_time"], ri["start_time"]), "nanostats start times"
    if "pair" not in SKIP:
        for gk, rk in (("q1", "q"), ("q2", "q2")):
            g, r = pair[gk], ref[rk]
            assert (g.number_of_reads, g.max_length) == (r.number_of_reads, r.max_length), ("paired", gk, "counters")
            for name in ("base_count_table", "phred_count_table", "end_anchored_base_count_table",
                         "end_anchored_phred_count_table", "gc_content", "phred_scores"):
                assert np.array_equal(u64(getattr(g, name)()), getattr(r, name)()), ("paired", gk, name)
        for gk, rk in (("p1", "p"), ("p2", "p2")):
            assert pair[gk].number_of_reads == ref[rk].number_of_reads, ("paired", gk, "reads", pair[gk].number_of_reads, ref[rk].number_of_reads)
            gt, rt = pair[gk].get_tile_counts(), ref[rk].get_tile_counts()
            assert len(gt) == len(rt), ("paired", gk, "tiles")
            for (t, e, c), (tr, er, cr) in zip(gt, rt):
                assert t == tr and np.array_equal(u64(c), cr) and np.allclose(np.array(e), er, rtol=1e-6), ("paired", gk, "tile", t)
        assert np.array_equal(u64(pair["z"].insert_sizes()), ref["z"].insert_sizes()), ("paired", "insert sizes")
        for which in ("adapters_read1", "adapters_read2"):
            ga, ra, sa = getattr(pair["z"], which)(), getattr(ref["z"], which)(), getattr(got["z"], which)()
            assert ga == ra, ("paired", which, "paired:", ga[:5], "oracle:", ra[:5], "standalone:", sa[:5],
                              "counters", pair["z"].total_reads, pair["z"].number_of_adapters_read1, ref["z"].number_of_adapters_read1)


failures = 0
for it in range(iters):
    if only is not None and it != only:
        continue
    rng = np.random.default_rng(seed0 * 1000 + it)
    n = int(rng.choice([1, 63, 64, 65, 500, 3000, 4500, 6000, 70000], p=[.12, .12, .12, .12, .12, .12, .12, .12, .04]))   # (70000: the length-sorted route takes batches from 65536 reads on)
    max_len = int(rng.choice([5, 40, 151, 300, 700, 2500]))
    if n * max_len > 16_000_000:
        n = 16_000_000 // max_len
    uniform, illumina = bool(rng.random() < 0.4), bool(rng.random() < 0.7)
    adapters = ADS[int(rng.integers(0, len(ADS)))]
    cuts = sorted({0, n, *(int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3))))})
    b1, m1 = make(rng, n, max_len, uniform, illumina, adapters)
    b2, m2 = make(rng, n, max_len, uniform, illumina, adapters)
    okw = dict(max_unique_fragments=int(rng.choice([50, 700, 5_000_000])), sample_every=int(rng.choice([1, 3, 8])),
               fragment_length=int(rng.choice([5, 21, 31])))
    dkw = dict(max_stored_fingerprints=int(rng.choice([100, 300, 1_000_000])),
               front_sequence_offset=int(rng.choice([0, 8, 64])), back_sequence_offset=int(rng.choice([0, 8])))
    zcap = int(rng.choice([3, 50, 10000]))
    ea = int(rng.choice([0, 7, 100, 300]))
    ref = dict(q=oracle.QCMetrics(ea), a=oracle.AdapterCounter(adapters), p=oracle.PerTileQuality(),
               o=oracle.OverrepresentedSequences(**okw), d=oracle.DedupEstimator(**dkw), dp=oracle.DedupEstimator(**dkw),
               z=oracle.InsertSizeMetrics(zcap), n=oracle.NanoStats())
    got = dict(q=QCMetrics(ea), a=AdapterCounter(adapters), p=PerTileQuality(), o=OverrepresentedSequences(**okw),
               d=DedupEstimator(**dkw), dp=DedupEstimator(**dkw), z=InsertSizeMetrics(zcap), n=NanoStats())
    fused = FusedPass(got["q"], got["a"], got["p"]) if rng.random() < 0.5 else None
    # the driver's call for paired input (PairedPass = the five calls per pair of arrays, __main__.py:279-306), objects of its own
    ref.update(q2=oracle.QCMetrics(ea), p2=oracle.PerTileQuality())
    pair = dict(q1=QCMetrics(ea), p1=PerTileQuality(), q2=QCMetrics(ea), p2=PerTileQuality(), z=InsertSizeMetrics(zcap))
    paired = PairedPass(pair["q1"], pair["p1"], pair["q2"], pair["p2"], pair["z"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            x1, x2 = m1[lo:hi].copy(), m2[lo:hi].copy()
            ref["q"].add(b1, x1); ref["a"].add(b1, x1); ref["p"].add(b1, x1); ref["o"].add(b1, x1)
            ref["d"].add(b1, x1); ref["dp"].add_pair(b1, x1, b2, x2); ref["z"].add_pair(b1, x1, b2, x2); ref["n"].add(b1, x1)
            ref["q2"].add(b2, x2); ref["p2"].add(b2, x2)
            a1 = FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy())
            a2 = FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy())
            if fused:
                fused.add_record_array(a1)
            else:
                got["q"].add_record_array(a1); got["a"].add_record_array(a1); got["p"].add_record_array(a1)
            if "o" not in SKIP: got["o"].add_record_array(a1)
            if "d" not in SKIP: got["d"].add_record_array(a1)
            if "dp" not in SKIP: got["dp"].add_record_array_pair(a1, a2)
            if "z" not in SKIP: got["z"].add_record_array_pair(a1, a2)
            if "n" not in SKIP: got["n"].add_record_array(a1)
            if "pair" not in SKIP:
                paired.add_record_array_pair(FastqRecordArrayView._from_buffer(b1, m1[lo:hi].copy()), FastqRecordArrayView._from_buffer(b2, m2[lo:hi].copy()))
    try:
        check(got, pair, ref, "arrays")
        if n <= 6000 and "parser" not in SKIP:
            # the same records once more, as the reference's driver feeds them (__main__.py:279-306): two FastqParsers over the
            # text (a random buffer size), read 2's arrays by read(len(array of read 1)), every module called per array
            import io
            from sequali_amd import FastqParser
            bs = int(rng.choice([3000, 20000, 128 * 1024, 1 << 20]))
            got2 = dict(q=QCMetrics(ea), a=AdapterCounter(adapters), p=PerTileQuality(), o=OverrepresentedSequences(**okw),
                        d=DedupEstimator(**dkw), dp=DedupEstimator(**dkw), z=InsertSizeMetrics(zcap), n=NanoStats())
            fused2 = FusedPass(got2["q"], got2["a"], got2["p"]) if fused else None
            pair2 = dict(q1=QCMetrics(ea), p1=PerTileQuality(), q2=QCMetrics(ea), p2=PerTileQuality(), z=InsertSizeMetrics(zcap))
            paired2 = PairedPass(pair2["q1"], pair2["p1"], pair2["q2"], pair2["p2"], pair2["z"])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r2 = FastqParser(io.BytesIO(b2), bs)
                r1b, r2b = FastqParser(io.BytesIO(b1), bs), FastqParser(io.BytesIO(b2), bs)
                for arr1, arr1b in zip(FastqParser(io.BytesIO(b1), bs), r1b):
                    arr2, arr2b = r2.read(len(arr1)), r2b.read(len(arr1b))
                    if fused2:
                        fused2.add_record_array(arr1)
                    else:
                        got2["q"].add_record_array(arr1); got2["a"].add_record_array(arr1); got2["p"].add_record_array(arr1)
                    if "o" not in SKIP: got2["o"].add_record_array(arr1)
                    if "d" not in SKIP: got2["d"].add_record_array(arr1)
                    if "dp" not in SKIP: got2["dp"].add_record_array_pair(arr1, arr2)
                    if "z" not in SKIP: got2["z"].add_record_array_pair(arr1, arr2)
                    if "n" not in SKIP: got2["n"].add_record_array(arr1)
                    if "pair" not in SKIP: paired2.add_record_array_pair(arr1b, arr2b)
            check(got2, pair2, ref, f"parser, buffers of {bs}")
        print(f"[{it}] ok  n={n} max_len={max_len} uniform={uniform} cuts={cuts} fused={bool(fused)}", flush=True)
    except AssertionError as e:
        failures += 1
        print(f"[{it}] FAIL {e!r} n={n} max_len={max_len} uniform={uniform} illumina={illumina} cuts={cuts} "
              f"okw={okw} dkw={dkw} zcap={zcap} ea={ea} fused={bool(fused)} adapters={adapters}", flush=True)
print("failures:", failures)
sys.exit(1 if failures else 0)
