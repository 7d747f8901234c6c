"""NanoStats (_qcmodule.c:4804-5430) -- the oracle's restatement against the reference's own object (oracle/_ref) on
nanopore headers and BAM tags nobody chose: times with and without fractions and zone, every month and leap years, channels,
missing and malformed fields (the module then stops and says why), duration / channel / parent tags of every integer
type.  CPU only; the GPU test feeds the same inputs to sequali_amd.NanoStats.  Skipped where oracle/_ref is absent."""
import io
import struct
import warnings

import numpy as np
import pytest

from oracle import oracle
from tests.test_oracle_vs_reference import REF, fastq

pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")


def timestamp(rng, how):
    y, mo, d = int(rng.integers(1970, 2100)), int(rng.integers(1, 13)), int(rng.integers(1, 29))
    if rng.random() < 0.2:
        mo, d = 2, 29 if (y % 4 == 0 and (y % 100 != 0 or y % 400 == 0)) else 28
    if rng.random() < 0.1:
        mo, d = int(rng.choice([1, 3, 5, 7, 8, 10, 12])), 31
    h, mi, s = int(rng.integers(0, 24)), int(rng.integers(0, 60)), int(rng.integers(0, 60))
    t = f"{y:04d}-{mo:02d}-{d:02d}T{h:02d}:{mi:02d}:{s:02d}"
    return t + {0: "Z", 1: ".123456Z", 2: "+00:00", 3: ".5+00:00", 4: "+02:00", 5: "", 6: ".Z", 7: "z", 8: ".123456789012+00:00"}[how]


def header(rng, i):
    kind = rng.random()
    how = int(rng.choice([0, 0, 0, 1, 2, 3])) if kind < 0.93 else int(rng.integers(4, 9))
    parts = [f"{i:08x}-aaaa-bbbb-cccc-{int(rng.integers(0, 1 << 40)):012x}", f"runid={int(rng.integers(0, 1 << 60)):x}", f"read={i}"]
    ch = f"ch={int(rng.integers(0, 5000))}"
    st = "start_time=" + timestamp(rng, how)
    extra = [f"flow_cell_id=FAK{i}", "protocol_group_id=x", "sample_id=s", "barcode=barcode01", "parent_read_id=" + parts[0]]
    fields = [ch, st] + [extra[int(k)] for k in rng.choice(len(extra), size=int(rng.integers(0, 3)), replace=False)]
    rng.shuffle(fields)
    if kind > 0.97:
        fields = [f for f in fields if not f.startswith("ch=")] if rng.random() < 0.5 else [f for f in fields if not f.startswith("start_time=")]
    if 0.95 < kind <= 0.97:
        fields = [f.replace("ch=", "ch=x") for f in fields]
    return " ".join(parts + fields)


def infos_of_reference(ns):
    return [(i.start_time, i.channel_id, i.length, i.parent_id_hash, np.float32(i.duration).view(np.uint32).item(),
             np.float64(i.cumulative_error_rate).view(np.uint64).item()) for i in ns.nano_info_iterator()]


def infos_of_oracle(ns):
    a = ns.nano_infos()
    return [(int(r["start_time"]), int(r["channel_id"]), int(r["length"]), int(r["parent_id_hash"]),
             int(r["duration"].view(np.uint32)), int(r["cumulative_error_rate"].view(np.uint64))) for r in a]


def fastq_case(seed):
    rng = np.random.default_rng(41000 + seed)
    n = int(rng.choice([1, 5, 60]))
    names = [header(rng, i) for i in range(n)]
    seqs, quals = [], []
    for _ in range(n):
        L = int(rng.integers(0, 400))
        seqs.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L).tobytes().decode())
        quals.append((rng.integers(0, 60, size=L) + 33).astype(np.uint8).tobytes().decode())
    return names, seqs, quals


@pytest.mark.parametrize("seed", range(80))
def test_fastq_headers(seed):
    names, seqs, quals = fastq_case(seed)
    arrays = list(REF.FastqParser(io.BytesIO(fastq(names, seqs, quals)), 1 << 22))
    rq, rn = REF.QCMetrics(), REF.NanoStats()
    for arr in arrays:
        rq.add_record_array(arr)     # writes accumulated_error_rate into the array (:2126), NanoStats reads it (:5314)
        rn.add_record_array(arr)
    buf, metas = oracle.make_batch(names, seqs, quals)
    oracle.QCMetrics().add(buf, metas)
    gn = oracle.NanoStats()
    gn.add(buf, metas)
    assert gn.number_of_reads == rn.number_of_reads
    assert (gn.minimum_time, gn.maximum_time) == (rn.minimum_time, rn.maximum_time)
    assert gn.skipped == (rn.skipped_reason is not None)
    assert infos_of_oracle(gn) == infos_of_reference(rn)


def bam_with_tags(rng, n):
    out = [b"BAM\x01", struct.pack("<I", 0), struct.pack("<I", 0)]
    for i in range(n):
        L = int(rng.integers(0, 200))
        name = (f"read{i}").encode() + b"\x00"
        tags = b""
        order = ["st", "du", "ch", "pi", "xx", "ar"]
        rng.shuffle(order)
        for t in order[:int(rng.integers(0, 7))]:
            if t == "st":
                tags += b"stZ" + timestamp(rng, int(rng.choice([0, 1, 2, 3]))).encode() + b"\x00"
            elif t == "du":
                tags += b"duf" + struct.pack("<f", float(rng.random() * 1000))
            elif t == "ch":
                code, fmt, lo, hi = [(b"c", "<b", -128, 128), (b"C", "<B", 0, 256), (b"s", "<h", -3000, 3000), (b"S", "<H", 0, 60000),
                                     (b"i", "<i", -5, 100000), (b"I", "<I", 0, 1 << 31)][int(rng.integers(0, 6))]
                tags += b"ch" + code + struct.pack(fmt, int(rng.integers(lo, hi)))
            elif t == "pi":
                tags += b"piZ" + f"{int(rng.integers(0, 1 << 60)):x}".encode() + b"\x00"
            elif t == "xx":
                tags += b"xxA" + b"q"
            else:
                sub, fmt, w = [(b"c", "<b", 1), (b"S", "<H", 2), (b"i", "<i", 4), (b"f", "<f", 4)][int(rng.integers(0, 4))]
                k = int(rng.integers(0, 5))
                vals = b"".join(struct.pack(fmt, 1.5 if fmt == "<f" else 1) for _ in range(k))
                tags += b"arB" + sub + struct.pack("<I", k) + vals
        r = rng.random()
        if r < 0.03:
            tags += b"stf" + struct.pack("<f", 1.0)       # the right tag with the wrong type (:5230-5245)
        elif r < 0.06:
            tags += b"duZ12\x00"
        elif r < 0.08:
            tags += b"zzQ" + b"\x01\x02"                    # a type that does not exist
        elif r < 0.10:
            tags += b"arBQ" + struct.pack("<I", 2) + b"\x00" * 8   # an array of a type that does not exist
        elif r < 0.12 and tags:
            tags = tags[:-1]                              # cut short
        seq = bytes(rng.integers(0, 256, size=(L + 1) // 2).astype(np.uint8))
        qual = bytes(rng.integers(0, 60, size=L).astype(np.uint8))
        body = struct.pack("<iiBBHHHIiii", -1, -1, len(name), 0, 4680, 0, 4, L, -1, -1, 0) + name + seq + qual + tags
        out += [struct.pack("<I", len(body)), body]
    return b"".join(out)


@pytest.mark.parametrize("seed", range(120))
def test_bam_tags(seed):
    rng = np.random.default_rng(42000 + seed)
    stream = bam_with_tags(rng, int(rng.choice([1, 6, 50])))
    rq, rn = REF.QCMetrics(), REF.NanoStats()
    ref_error = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            for arr in REF.BamParser(io.BytesIO(stream), 1 << 22):
                rq.add_record_array(arr)
                rn.add_record_array(arr)
        except (ValueError, RuntimeError) as e:
            ref_error = type(e).__name__
    out, metas, consumed, _ = oracle.bam_decode(stream[12:])
    oracle.QCMetrics().add(out, metas)
    gn = oracle.NanoStats()
    got_error = None
    try:
        gn.add(out, metas)
    except oracle.NanoStatsError as e:
        got_error = {1: "ValueError", 2: "ValueError", 3: "ValueError", 4: "RuntimeError", 5: "SystemError"}[e.code]
    assert got_error == ref_error
    assert gn.number_of_reads == rn.number_of_reads
    assert (gn.minimum_time, gn.maximum_time) == (rn.minimum_time, rn.maximum_time)
    assert gn.skipped == (rn.skipped_reason is not None)
    assert infos_of_oracle(gn) == infos_of_reference(rn)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(30))
def test_gpu_nanostats_on_the_same_headers(seed):
    from sequali_amd import FastqParser, NanoStats, QCMetrics
    names, seqs, quals = fastq_case(seed)
    text = fastq(names, seqs, quals)
    rq, rn = REF.QCMetrics(), REF.NanoStats()
    for arr in REF.FastqParser(io.BytesIO(text), 1 << 22):
        rq.add_record_array(arr)
        rn.add_record_array(arr)
    gq, gn = QCMetrics(), NanoStats()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for arr in FastqParser(io.BytesIO(text), 1 << 22):
            gq.add_record_array(arr)
            gn.add_record_array(arr)
        gn.flush()
    assert gn.number_of_reads == rn.number_of_reads
    assert (gn.minimum_time, gn.maximum_time) == (rn.minimum_time, rn.maximum_time)
    assert gn.skipped_reason == rn.skipped_reason
    assert infos_of_reference(gn) == infos_of_reference(rn)
