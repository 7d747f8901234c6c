"""CPU-only tests of the drop-in boundary: the C ABI library loads and exports
every symbol of include/sqgpu.h, the host-side record types behave like the
reference's (golden vectors / oracle), the synthetic generator is deterministic.
No GPU work is launched here."""
import io
import re

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import golden, golden_json, split_fastq


def test_library_exports_every_declared_symbol():
    from sequali_amd import _lib
    lib = _lib.lib()
    header = open(_lib.HEADER).read()
    declared = set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 85
    assert declared == set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sq_abi_version() == 1


def test_init_without_gpu_fails_loudly():
    """no silent CPU fallback: without a device sq_init reports an error"""
    import torch
    from sequali_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert not _lib.lib().sq_init(0)
    assert "hipGetDeviceCount" in _lib.last_error() or "not available" in _lib.last_error()


def test_meta_layout_matches_fastqmeta():
    from sequali_amd._qc import META_DTYPE
    assert META_DTYPE.itemsize == 40
    assert [META_DTYPE.fields[n][1] for n in META_DTYPE.names] == [0, 8, 12, 16, 20, 24, 28, 32]


def test_constants():
    import sequali_amd as s
    assert (s.NUMBER_OF_NUCS, s.NUMBER_OF_PHREDS, s.TABLE_SIZE, s.PHRED_MAX) == (5, 12, 60, 93)
    assert (s.A, s.C, s.G, s.T, s.N) == (0, 1, 2, 3, 4)
    assert s.MAX_SEQUENCE_SIZE == 64 and s.INSERT_SIZE_MAX_ADAPTER_STORE_SIZE == 31
    assert s.DEFAULT_MAX_UNIQUE_FRAGMENTS == 5_000_000 and s.DEFAULT_FRAGMENT_LENGTH == 21


def test_record_view_roundtrip_and_error_rate():
    from sequali_amd import FastqRecordView
    v = FastqRecordView("name 1", "ACGTN", "!+5?I", b"tags")
    assert (v.name(), v.sequence(), v.qualities(), v.tags()) == ("name 1", "ACGTN", "!+5?I", b"tags")
    want = 0.0
    for c in "!+5?I":
        want += oracle.error_rate(ord(c) - 33)
    assert v._meta["accumulated_error_rate"][0] == want
    assert v.obj == b"name 1ACGTN!+5?Itags"


@pytest.mark.parametrize("args,exc,match", [
    (("n", "ACGT", "III"), ValueError, "different lengths: 4 and 3"),
    (("n", "ACGT", "II\x7fI"), ValueError, "Not a valid phred character: \x7f"),
    (("n", "ACGT", "II I"), ValueError, "Not a valid phred character:  "),
    (("nä", "A", "I"), ValueError, "name should contain only ASCII"),
    (("n", "ä", "I"), ValueError, "sequence should contain only ASCII"),
    ((b"n", "A", "I"), TypeError, "must be str"),
])
def test_record_view_errors(args, exc, match):
    from sequali_amd import FastqRecordView
    with pytest.raises(exc, match=match):
        FastqRecordView(*args)


def test_record_array_from_views_points_at_own_buffer():
    from sequali_amd import FastqRecordArrayView, FastqRecordView
    views = [FastqRecordView(f"r{i}", "ACGT" * i, "IIII" * i) for i in range(5)]
    arr = FastqRecordArrayView(views)
    assert len(arr) == 5
    for i in range(5):
        assert arr[i].name() == f"r{i}" and arr[i].sequence() == "ACGT" * i
        assert arr[i].obj is arr.obj   # not the source view's buffer
    assert arr[-1].qualities() == "IIII" * 4
    with pytest.raises(IndexError):
        arr[5]
    with pytest.raises(TypeError, match="FastqRecordView"):
        FastqRecordArrayView([views[0], "no"])


def test_is_mate_golden_and_errors():
    from sequali_amd import FastqRecordArrayView, FastqRecordView
    for a, b, want in golden_json("is_mate"):
        x = FastqRecordArrayView([FastqRecordView(a, "A", "A")])
        y = FastqRecordArrayView([FastqRecordView(b, "A", "A")])
        assert x.is_mate(y) is want
        assert oracle.names_are_mates(a, b) is want
    with pytest.raises(TypeError, match="FastqRecordArrayView"):
        x.is_mate("nope")
    with pytest.raises(ValueError, match="same length"):
        x.is_mate(FastqRecordArrayView([]))


@pytest.mark.parametrize("name", ["ref_simple", "ref_100_illumina_adapters", "ref_100_nanopore"])
@pytest.mark.parametrize("buffersize", [1, 7, 100, 4096, 128 * 1024, 1 << 24])
def test_fastq_parser_chunking(name, buffersize):
    """tests/test_fastq_parser.py:49-174 of the reference: any buffer size gives the
    same records; offsets follow FastqParser's layout"""
    from sequali_amd import FastqParser
    text = golden(name)["fastq"].tobytes()
    if buffersize < 100 and len(text) > 100_000:
        pytest.skip("quadratic for a tiny buffer on a large file")
    buf, metas = split_fastq(text)
    got = []
    for arr in FastqParser(io.BytesIO(text), buffersize):
        assert len(arr) > 0
        for i in range(len(arr)):
            v = arr[i]
            got.append((v.name(), v.sequence(), v.qualities()))
    assert len(got) == len(metas)
    for (n, s, q), m in zip(got, metas):
        st = int(m["record_start"])
        assert n == text[st:st + int(m["name_length"])].decode()
        assert s == text[st + int(m["sequence_offset"]):st + int(m["sequence_offset"]) + int(m["sequence_length"])].decode()
        assert q == text[st + int(m["qualities_offset"]):st + int(m["qualities_offset"]) + int(m["sequence_length"])].decode()


def test_fastq_parser_read_lockstep_and_errors():
    from sequali_amd import FastqParser
    text = golden("ref_100_illumina_adapters")["fastq"].tobytes()
    p = FastqParser(io.BytesIO(text), 500)
    sizes = []
    while True:
        arr = p.read(7)
        if len(arr) == 0:
            break
        sizes.append(len(arr))
    assert sum(sizes) == 100 and all(s == 7 for s in sizes[:-1])
    assert list(FastqParser(io.BytesIO(b""))) == []
    with pytest.raises(EOFError, match="Incomplete record"):
        list(FastqParser(io.BytesIO(b"@a\nACGT\n+\nII")))
    with pytest.raises(ValueError, match="does not start with @"):
        list(FastqParser(io.BytesIO(b"a\nACGT\n+\nIIII\n")))
    with pytest.raises(ValueError, match="does not start with \\+"):
        list(FastqParser(io.BytesIO(b"@a\nACGT\n-\nIIII\n")))
    with pytest.raises(ValueError, match="equal length"):
        list(FastqParser(io.BytesIO(b"@a\nACGT\n+\nIII\n")))
    with pytest.raises(ValueError, match="non-ASCII"):
        list(FastqParser(io.BytesIO("@a\nACGä\n+\nIIII\n".encode("latin-1"))))
    with pytest.raises(ValueError):
        FastqParser(io.BytesIO(b""), 0)


def test_synthetic_generator_is_counter_based():
    """any slice of the stream can be generated on its own (what lets every rank
    generate its shard) and the text parses back to the same metas"""
    from sequali_amd import synth
    whole, metas = synth.host_records(synth.ILLUMINA, 0, 300)
    part, _ = synth.host_records(synth.ILLUMINA, 100, 50)
    assert whole[100 * 348:150 * 348] == part
    buf, parsed = split_fastq(whole)
    for f in ("record_start", "name_length", "sequence_offset", "sequence_length", "qualities_offset"):
        np.testing.assert_array_equal(parsed[f], metas[f])
    assert len(set(whole[i * 348:(i + 1) * 348] for i in range(300))) == 300
    r1, _ = synth.host_records(synth.ILLUMINA, 0, 200)
    r2, m2 = synth.host_records(synth.ILLUMINA_R2, 0, 200)
    n1 = [r1[i * 348 + 1:i * 348 + 30] for i in range(200)]
    n2 = [r2[i * 348 + 1:i * 348 + 30] for i in range(200)]
    assert n1 == n2    # mates share the id part of the header
    nano, mn = synth.host_records(synth.NANOPORE, 5, 40)
    assert 200 <= mn["sequence_length"].min() and mn["sequence_length"].max() <= 100000
    assert len(np.unique(mn["sequence_length"])) > 30
    _, again = synth.host_records(synth.NANOPORE, 5, 40)
    np.testing.assert_array_equal(mn, again)


def test_synthetic_data_has_the_intended_features():
    """duplicates, adapter read-through, N bases, four quality values, 96 tiles"""
    from sequali_amd import synth
    text, metas = synth.host_records(synth.ILLUMINA, 0, 5000)
    seqs = [text[int(m["record_start"]) + int(m["sequence_offset"]):][:150] for m in metas]
    assert 0.03 < sum(b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"[:20] in s for s in seqs) / 5000 < 0.15
    assert 4000 < len(set(s[20:60] for s in seqs)) < 4900     # ~10 % duplicated fragments
    allbytes = b"".join(seqs)
    assert 0.0005 < allbytes.count(b"N") / len(allbytes) < 0.002
    quals = set(b"".join(text[int(m["record_start"]) + int(m["qualities_offset"]):][:150] for m in metas[:200]))
    assert quals == set(b"F:,#")
    tiles = set(oracle.tile_id(text[int(m["record_start"]):][:int(m["name_length"])].decode()) for m in metas)
    assert len(tiles) == 96 and min(tiles) == 1101 and max(tiles) == 2224


# ---- the adapter automatons (host tables, no GPU) ---------------------------------------
def _automaton_tables(adapters):
    import ctypes as C
    from sequali_amd._lib import lib
    enc = [a.encode("ascii") for a in adapters]
    arr = (C.c_char_p * len(enc))(*enc)
    lens = (C.c_size_t * len(enc))(*[len(e) for e in enc])
    cap = 4096
    dfa, out = np.zeros(cap * 8, np.uint16), np.zeros(cap, np.uint64)
    dfa2, out2 = np.zeros(cap * 36, np.uint16), np.zeros(cap * 2, np.uint64)
    accept, states2 = C.c_uint32(0), C.c_uint32(0)
    n = lib().sq_adapter_automaton_tables(arr, lens, len(enc), dfa.ctypes.data, out.ctypes.data, cap, C.byref(accept),
                                          dfa2.ctypes.data, out2.ctypes.data, cap, C.byref(states2))
    assert n > 0
    return (dfa[:n * 8].reshape(n, 8), out[:n], int(accept.value),
            dfa2[:states2.value * 36].reshape(-1, 36), out2[:states2.value * 2].reshape(-1, 2))


def _classes(text: bytes):
    return [{"a": 0, "c": 1, "g": 2, "t": 3}.get(chr(b).lower(), 4) for b in text]


@pytest.mark.parametrize("adapters", [
    ["AGATCGGAAGAG", "TGGAATTCTCGG", "GATCGTCGGACT", "CTGTCTCTTATA", "GGGGGGGGGGGG", "AAAAAAAAAAAA"],
    ["ACGT", "CGTA", "GT", "A", "TTTT", "ACGTACGTACGTA"],
    ["GGGG", "GGGGG", "GG", "ANA", "NN"],
])
def test_pair_automaton_reports_what_the_single_step_one_reports(adapters):
    """k_span walks the automaton two characters per step (build_pair_dfa): on random text, from
    every phase, it reports the same (adapter, end position) events as the one-character automaton
    -- itself checked here against a plain search -- except for what the silent twins swallow in
    both: repeats of an adapter inside a run of its own characters, which AdapterCounter never
    counts twice in a read anyway (update_adapter_count_array, _qcmodule.c:2643-2672: first hit per
    adapter and read).  So the comparison is on first hits."""
    dfa, out, accept, dfa2, out2 = _automaton_tables(adapters)
    assert (np.nonzero(out)[0] >= accept).all() and len(dfa2) >= len(dfa)
    assert (out2[:len(out), 0] == out).all() and (out2[:len(out), 1] == 0).all()
    rng = np.random.default_rng(len(adapters[0]))
    for trial in range(300):
        L = int(rng.integers(1, 90))
        text = rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), size=L, p=[.2, .2, .2, .2, .02, .04, .04, .04, .04, .02]).tobytes()
        if trial % 3 == 0:   # plant adapters and runs
            for _ in range(3):
                w = adapters[int(rng.integers(0, len(adapters)))].encode()
                if len(w) <= L:
                    at = int(rng.integers(0, L - len(w) + 1))
                    text = text[:at] + w + text[at + len(w):]
        if trial % 5 == 0:
            text = text[:L // 2] + b"G" * (L - L // 2)
        cls = _classes(text)
        # plain search: first end position of every adapter (N in an adapter matches class 4, :2451-2462)
        want = {}
        for a, ad in enumerate(adapters):
            ac = _classes(ad.encode())
            for e in range(len(ac) - 1, L):
                if cls[e - len(ac) + 1:e + 1] == ac:
                    want[a] = e
                    break
        def first_hits(events):
            first = {}
            for a, e in events:
                first[a] = min(first.get(a, e), e)
            return first
        # one character per step
        s, ev1 = 0, []
        for i, c in enumerate(cls):
            s = int(dfa[s, c]) >> 4
            ev1 += [(a, i) for a in range(len(adapters)) if int(out[s]) >> a & 1]
        assert first_hits(ev1) == want
        # two characters per step, from both phases (the second one starts with a lone first character:
        # (padding, c) from the root is a one-character step)
        for phase in (0, 1):
            seq = ([5] if phase else []) + cls
            if len(seq) % 2:
                seq = seq + [5]
            s, ev2 = 0, []
            for i in range(0, len(seq), 2):
                s = int(dfa2[s, seq[i] + 6 * seq[i + 1]])
                pos = i + 1 - phase   # position of the step's second character in the text
                ev2 += [(a, pos) for a in range(len(adapters)) if int(out2[s, 0]) >> a & 1]
                ev2 += [(a, pos - 1) for a in range(len(adapters)) if int(out2[s, 1]) >> a & 1]
                assert (s >= accept) == bool(out2[s, 0] or out2[s, 1])
            assert first_hits(ev2) == want, (text, phase)


# ---- the parser's buffer logic in the C ABI (sq_feeder) against the Python restatement -----------
def _parse_all(text, buffersize, reads=None):
    """(sizes, buffer lengths, metas, error) of a FastqParser run over `text`"""
    from sequali_amd import FastqParser
    p = FastqParser(io.BytesIO(text), buffersize)
    sizes, lens, metas, err = [], [], [], None
    try:
        if reads is None:
            arrays = list(p)
        else:
            arrays = []
            for n in reads:
                a = p.read(n) if n else next(p, None)
                if a is None or len(a) == 0:
                    break
                arrays.append(a)
        for a in arrays:
            sizes.append(len(a))
            lens.append(len(a.obj))
            m = a._host_metas()
            metas.append(np.stack([m[f].astype(np.int64) for f in ("record_start", "name_length", "sequence_offset",
                                                                    "sequence_length", "qualities_offset", "tags_offset")], axis=1))
            for i in (0, len(a) - 1):
                r = a[i]
                assert len(r.sequence()) == len(r.qualities())
    except (ValueError, EOFError) as e:
        err = (type(e).__name__, str(e))
    return sizes, lens, (np.concatenate(metas).tolist() if metas else []), err


@pytest.mark.parametrize("seed", range(12))
def test_feeder_equals_the_python_parser_on_random_texts(seed):
    """sq_feeder (pinned staging blocks, vectorised split, read-ahead) and the Python restatement of
    FastqParser_create_record_array (_qcmodule.c:964-1184) that the golden parser cases pinned in
    round 2: same arrays, same buffer lengths, same metas, same exception -- records of very
    different sizes (some larger than the buffer: it is enlarged), buffer sizes from 1 byte up,
    truncated tails, a broken record, a non-ASCII byte, iteration mixed with read(n)"""
    from sequali_amd import _qc
    rng = np.random.default_rng(900 + seed)
    recs = []
    for i in range(int(rng.integers(1, 60))):
        L = int(rng.choice([0, 1, 5, 40, 150, 700, 5000], p=[.05, .1, .2, .25, .25, .1, .05]))
        name = "r%d %s" % (i, "x" * int(rng.integers(0, 30)))
        seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L).tobytes()
        qual = (rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes()
        recs.append(b"@" + name.encode() + b"\n" + seq + b"\n+\n" + qual + b"\n")
    text = b"".join(recs)
    variants = [text, text[:max(1, len(text) - int(rng.integers(1, 40)))]]
    if len(recs) > 2:
        bad = list(recs)
        k = int(rng.integers(1, len(bad)))
        bad[k] = bad[k].replace(b"\n+\n", b"\n-\n", 1) if seed % 2 else b"X" + bad[k][1:]
        variants.append(b"".join(bad))
        hi = bytearray(text)
        hi[int(rng.integers(0, len(hi)))] = 0xC3
        variants.append(bytes(hi))
    for t in variants:
        for bs in (1, 7, 64, 333, 4096, 128 * 1024):
            for reads in (None, [3, 0, 1, 0, 0, 5, 2, 0] * 20):
                got = _parse_all(t, bs, reads)
                _qc._USE_FEEDER = False
                try:
                    want = _parse_all(t, bs, reads)
                finally:
                    _qc._USE_FEEDER = True
                assert got == want, (seed, bs, reads is not None, got[3], want[3])


def test_pinned_reader_is_a_file_object():
    from sequali_amd import FastqParser, PinnedReader
    text = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(1000))
    r = PinnedReader(text)
    assert r.read(5) == text[:5]
    buf = bytearray(11)
    assert r.readinto(buf) == 11 and bytes(buf) == text[5:16]
    assert r.read() == text[16:] and r.read(3) == b"" and r.readinto(bytearray(4)) == 0
    assert r.tell() == len(text) and r.seek(7) == 7 and r.read(4) == text[7:11] and r.seek(-3, 2) == len(text) - 3
    assert r.read() == text[-3:] and r.seek(0) == 0
    arrays = list(FastqParser(PinnedReader(text), 300))        # through the host parser like any file
    assert sum(len(a) for a in arrays) == 1000 and arrays[0][0].name() == "r0"


def test_sum_thresholds_say_what_the_division_says():
    """k_span compares a read's SUM of error rates with sums[k] instead of sum / length with thresholds[k]
    (DESIGN 4.1b): sums[k] must be the largest double whose quotient is still <= thresholds[k], for every length"""
    import ctypes as C
    import math
    import numpy as np
    from sequali_amd._lib import lib
    thr = (C.c_double * 94)()
    sums = (C.c_double * 94)()
    for length in list(range(1, 257)):
        lib().sq_phred_sum_thresholds(length, thr, sums)
        t, s = np.array(thr[:]), np.array(sums[:])
        assert math.isinf(t[0]) and math.isinf(s[0])
        d = np.float64(length)
        assert np.all(s[1:] / d <= t[1:]), length
        assert np.all(np.nextafter(s[1:], np.inf) / d > t[1:]), length
        assert np.all(np.diff(s[1:]) < 0)          # falls with k like the thresholds
    # the bins of sums around every threshold, by the reference's formula (floor(-10 log10(avg)), :2127-2136)
    lib().sq_phred_sum_thresholds(150, thr, sums)
    for k in range(1, 94):
        for x in (sums[k], float(np.nextafter(sums[k], np.inf))):
            want = int(math.floor(-10.0 * math.log10(x / 150.0)))
            got = max(i for i in range(94) if x <= sums[i])
            assert got == min(want, 93), (k, x)


def test_a_parser_that_has_ended_gives_its_blocks_back():
    """the staging blocks of a FastqParser (page-locked on a GPU box: 6-9 ms per 64 MiB to lock) return to the
    process-wide pool when the parser and its arrays are gone -- without waiting for Python's cycle collector --
    so that the next parser allocates nothing (sq_feeder_debug_times counts fresh allocations)"""
    import ctypes as C
    import gc
    import io
    from sequali_amd import FastqParser
    from sequali_amd._lib import lib
    text = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(20000))
    out = (C.c_double * 4)()
    gc.disable()
    try:
        for rep in range(3):
            lib().sq_feeder_debug_times(out, 1)
            arrays = list(FastqParser(io.BytesIO(text), 4096))
            assert sum(len(a) for a in arrays) == 20000
            del arrays
            lib().sq_feeder_debug_times(out, 0)
            if rep:
                assert out[3] == 0, "a later parser had to allocate staging memory afresh"
    finally:
        gc.enable()


def test_workgroup_shares_of_equal_cost():
    """k_span over the segments of long reads: every span belongs to exactly one workgroup, the shares
    are consecutive, and no workgroup's cost (spans + 16 per segment it meets) is far from the mean -- with equal
    numbers of spans the last workgroup of config 4 went through 155 segments"""
    import ctypes as C
    import numpy as np
    from sequali_amd._lib import lib
    rng = np.random.default_rng(4)
    lengths = np.clip(rng.lognormal(np.log(8000), 0.75, 200_000), 200, 100_000).astype(np.int64)
    nj = [(lengths > 256 * j).sum() for j in range(int((lengths.max() + 255) // 256))]
    nspans = np.array([-(-int(n) // 16) for n in nj if n], dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(nspans)])
    for grid, cost in ((256, 16), (256, 1), (7, 16), (1, 16), (300, 48)):
        bounds = (C.c_uint32 * (grid + 1))()
        lib().sq_span_cost_shares(nspans.ctypes.data, len(nspans), grid, cost, bounds)
        b = np.array(bounds[:], dtype=np.int64)
        assert b[0] == 0 and b[-1] == nspans.sum() and np.all(np.diff(b) >= 0)
        costs = []
        for w in range(grid):
            lo, hi = b[w], b[w + 1]
            met = int(np.searchsorted(starts, hi, "left") - np.searchsorted(starts, lo, "right") + 1) if hi > lo else 0
            costs.append((hi - lo) + cost * met)
        if grid >= 7:
            assert max(costs) <= 1.05 * np.mean(costs) + 2 * cost, (grid, cost, max(costs), np.mean(costs))
    # equal numbers of spans, for comparison: the last share meets most of the segments
    per = -(-int(nspans.sum()) // 256)
    last_lo = 255 * per
    assert len(nspans) - np.searchsorted(starts, last_lo, "right") > 50


def test_rccl_entry_points_load_and_fail_loudly_without_a_communicator():
    """the C ABI's own exchange step (csrc/sq_dist.hip: RCCL opened on first use, no torch): the entry points exist,
    librccl.so and its symbols are found, and a call without a context or a communicator is an error with a message,
    not a crash (no N > 1 run exists yet: no node with more than one GPU has been available)"""
    import ctypes as C
    from sequali_amd._lib import last_error, lib
    L = lib()
    assert L.sq_rccl_available() == 1, last_error()
    ptrs = (C.c_void_p * 1)(None)
    counts = (C.c_uint64 * 1)(0)
    assert L.sq_rccl_allreduce_tables(None, None, ptrs, counts, 1, 0) < 0
    assert "communicator" in last_error()
    assert L.sq_rccl_allgather_bytes(None, None, None, None, 0) < 0
    assert L.sq_rccl_comm_init(None, 2, (C.c_uint8 * 128)(), 0) is None
    assert "context" in last_error()


def _reference_tile_id(name: bytes) -> int:
    """illumina_header_to_tile_id + unsigned_decimal_integer_from_string (_qcmodule.c:3088-3121, :159-180), as plain
    Python: the decimal number between the 4th and the 5th colon (at most 18 digits), -1 for anything else"""
    n, colons, cursor = len(name), 0, 0
    while cursor < n:
        if name[cursor] == 0x3A:
            colons += 1
            if colons == 4:
                break
        cursor += 1
    cursor += 1
    start = cursor
    while cursor < n:
        if name[cursor] == 0x3A:
            digits = name[start:cursor]
            if not 1 <= len(digits) <= 18 or any(not 0x30 <= c <= 0x39 for c in digits):
                return -1
            return int(digits)
        cursor += 1
    return -1


def test_header_parse_of_the_paired_pass_on_the_host():
    """k_span<PT> parses the tile id out of the first 64 bytes of a header held in registers (tile_id_of_words<8>) and
    falls back to a byte-by-byte walk (tile_id_of) for longer headers and tiles of 9 .. 18 digits; both are compiled for
    the host too (sq_test_tile_of_header).  Headers of every length from 0 to 100, the tile field at every offset (also
    across bytes 48 and 64), tiles of 0 to 20 digits, non-digits, too few colons; whatever lies behind the name (the
    kernel's loads fetch 64 bytes whatever the length) must not matter"""
    import ctypes as C
    from sequali_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(7)
    cases = [b"", b":", b"::::", b":::::", b"::::1:", b"::::12", b"a:b:c:d:007:x", b"a:b:c:d::x", b"a:b:c:d:12a:x",
             b"SIM:1:FCX:3:2221:22916:753886 1:N:0:ATCCGA", b"@:@:@:@:" + b"9" * 18 + b":", b"@:@:@:@:" + b"9" * 19 + b":",
             b"a:b:c:d:123456789:e", b"a:b:c:d:12345678:e"]
    for pad in range(0, 70):                       # the tile field slides over the word boundaries 8 .. 64
        for digits in (1, 4, 8, 9, 12):
            t = "".join(str(int(x)) for x in rng.integers(0, 10, size=digits))
            cases.append(b"I" * pad + f":1:F:2:{t}:55:66 1:N:0:AC".encode())
            cases.append(b"I" * pad + f":1:F:2:{t}".encode())             # no fifth colon
    for n in range(0, 101):
        raw = rng.choice(np.frombuffer(b"AC:019: x", np.uint8), size=n).tobytes()
        cases.append(raw)
    checked = 0
    for name in cases:
        for fill in (b"\xff", b":", b"7"):
            buf = (name + fill * 72)[:max(len(name), 64) + 8]
            got = L.sq_test_tile_of_header(C.c_char_p(buf), len(name))
            assert got == _reference_tile_id(name), (name, fill, got)
            got = L.sq_test_tile_of_header_quad(C.c_char_p(buf), len(name))     # the parse the four lanes of a quad share
            assert got == _reference_tile_id(name), ("quad", name, fill, got)
            checked += 1
    assert checked > 2000


def test_rccl_enum_values_of_sq_dist_are_those_of_rccl_h():
    """csrc/sq_dist.hip keeps its own copy of the five enum values it passes to RCCL (so that building libsqgpu.so
    needs no RCCL header): pinned here against the installed rccl.h and against the source's own text"""
    import os
    import re
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("rccl.h is not installed here")
    text = open(hdr).read()
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sequali_amd", "csrc", "sq_dist.hip")).read()
    for name in ("ncclSum", "ncclMax", "ncclUint8", "ncclUint64", "ncclFloat64", "ncclSuccess"):
        in_hdr = re.search(r"\b%s\s*=\s*(\d+)" % name, text)
        in_src = re.search(r"\b%s\s*=\s*(\d+)" % name, src)
        assert in_hdr and in_src, name
        assert int(in_hdr.group(1)) == int(in_src.group(1)), name
    assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128\b", text) and re.search(r"char\s+internal\[NCCL_UNIQUE_ID_BYTES\]", text)   # ncclUniqueId is the 128 bytes the ABI hands around
