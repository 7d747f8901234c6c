"""BAM input against the reference's own BamParser (oracle/_ref, _qcmodule.c:1266-1703) on random
uncompressed BAM streams nobody chose: records with and without qualities (0xff), odd sequence lengths (the last nibble),
every 4-bit base code, tags, secondary / supplementary alignments (skipped), reverse-strand flags, long read names.

CPU: the oracle's decode (oracle.bam_decode) and the host record walk of the product (sq_bam_scan).
GPU: sequali_amd.BamParser (walk on the host, decode on the GPU) -- oracle/_ref travels to the GPU box.
Skipped where oracle/_ref is absent."""
import ctypes as C
import io
import struct

import numpy as np
import pytest

from oracle import oracle
from tests.test_oracle_vs_reference import REF

pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")


def random_bam(rng, n, max_len):
    """(stream, number of records the parser hands out)"""
    text = b"@HD\tVN:1.6\tSO:unsorted\n" + (b"@CO\t" + bytes(rng.integers(32, 127, size=int(rng.integers(0, 50))).astype(np.uint8)) + b"\n" if rng.random() < 0.5 else b"")
    refs = [(b"chr%d\x00" % i, 1000 + i) for i in range(int(rng.integers(0, 3)))]
    out = [b"BAM\x01", struct.pack("<I", len(text)), text, struct.pack("<I", len(refs))]
    for name, ln in refs:
        out += [struct.pack("<I", len(name)), name, struct.pack("<I", ln)]
    kept = 0
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        name = bytes(rng.integers(33, 127, size=int(rng.integers(1, 60))).astype(np.uint8)) + b"\x00"
        flag = int(rng.choice([4, 77, 141, 4 | 16, 4 | 256, 4 | 2048, 4 | 512, 4 | 1024, 0]))
        n_cigar = int(rng.choice([0, 0, 0, 1, 3]))
        cigar = bytes(rng.integers(0, 256, size=4 * n_cigar).astype(np.uint8))
        seq = bytes(rng.integers(0, 256, size=(L + 1) // 2).astype(np.uint8))      # any nibble: =ACMGRSVTWYHKDBN
        if rng.random() < 0.2:
            qual = b"\xff" * L          # no qualities stored (:1640-1660)
        else:
            qual = bytes(rng.integers(0, 94, size=L).astype(np.uint8))
        tags = b""
        for _ in range(int(rng.integers(0, 4))):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                tags += b"RG" + b"Z" + bytes(rng.integers(33, 127, size=int(rng.integers(0, 12))).astype(np.uint8)) + b"\x00"
            elif kind == 1:
                tags += b"NM" + b"i" + struct.pack("<i", int(rng.integers(-5, 1000)))
            else:
                tags += b"du" + b"f" + struct.pack("<f", float(rng.random() * 100))
        body = struct.pack("<iiBBHHHIiii", -1, -1, len(name), int(rng.integers(0, 61)), 4680, n_cigar, flag, L, -1, -1, 0) + name + cigar + seq + qual + tags
        out += [struct.pack("<I", len(body)), body]
        kept += not (flag & (256 | 2048))
    return b"".join(out), kept


def reference_records(stream, buffer_size):
    """[(records of an array, [(name, sequence, qualities, tags)])] or what came before the error + (type, message)"""
    arrays = []
    try:
        for arr in REF.BamParser(io.BytesIO(stream), buffer_size):
            arrays.append([(arr[i].name(), arr[i].sequence(), arr[i].qualities(), arr[i].tags()) for i in range(len(arr))])
    except (ValueError, EOFError, OverflowError) as e:
        return arrays, (type(e).__name__, str(e))
    return arrays, None


def body_of(stream):
    l_text = struct.unpack_from("<I", stream, 4)[0]
    pos = 8 + l_text
    n_ref = struct.unpack_from("<I", stream, pos)[0]
    pos += 4
    for _ in range(n_ref):
        pos += 4 + struct.unpack_from("<I", stream, pos)[0] + 4
    return stream[pos:]


@pytest.mark.parametrize("seed", range(40))
def test_oracle_decode_and_host_walk(seed):
    from sequali_amd._lib import lib
    rng = np.random.default_rng(31000 + seed)
    stream, kept = random_bam(rng, int(rng.choice([0, 1, 9, 200])), int(rng.choice([0, 1, 7, 150, 3000])))
    want, err = reference_records(stream, 1 << 24)
    assert err is None and sum(len(a) for a in want) == kept
    want = [r for a in want for r in a]
    body = body_of(stream)
    out, metas, consumed, skipped = oracle.bam_decode(body)
    assert consumed == len(body) and len(metas) == kept
    got = []
    for m in metas:
        st, nl, so, sl, qo, to, tl = (int(m[f]) for f in ("record_start", "name_length", "sequence_offset", "sequence_length",
                                                           "qualities_offset", "tags_offset", "tags_length"))
        got.append((out[st:st + nl].decode("ascii"), out[st + so:st + so + sl].decode("ascii"),
                    out[st + qo:st + qo + sl].decode("ascii"), out[st + to:st + to + tl]))
    assert got == want
    view = np.frombuffer(body, dtype=np.uint8) if body else np.zeros(1, np.uint8)
    c, s = C.c_size_t(0), C.c_uint64(0)
    assert lib().sq_bam_scan(view.ctypes.data, len(body), None, 0, C.byref(c), C.byref(s)) == kept      # the product's walk
    assert (c.value, s.value) == (consumed, skipped)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_gpu_bam_parser(seed):
    """the product's BamParser: the same arrays (sizes by buffer size), the same records"""
    from sequali_amd import BamParser
    rng = np.random.default_rng(32000 + seed)
    stream, kept = random_bam(rng, int(rng.choice([1, 9, 200, 2000])), int(rng.choice([0, 7, 150, 3000])))
    for buffer_size in (1 << 24, 48 * 1024, 300):
        want, err = reference_records(stream, buffer_size)
        got = []
        try:
            for arr in BamParser(io.BytesIO(stream), buffer_size):
                got.append([(arr[i].name(), arr[i].sequence(), arr[i].qualities(), arr[i].tags()) for i in range(len(arr))])
            gerr = None
        except (ValueError, EOFError, OverflowError) as e:
            gerr = (type(e).__name__, str(e))
        assert gerr == err
        assert [len(a) for a in got] == [len(a) for a in want], buffer_size
        assert got == want


@pytest.mark.parametrize("seed", range(60))
def test_damaged_streams_on_the_host(seed):
    """a stream cut short or with a damaged header: the errors the parser raises before anything is decoded (no GPU),
    and for streams that still hold whole records, the same count from the walk"""
    from sequali_amd import BamParser
    rng = np.random.default_rng(33000 + seed)
    stream, _ = random_bam(rng, int(rng.choice([0, 1, 4])), 30)
    b = bytearray(stream)
    kind = seed % 3
    if kind == 0:
        del b[int(rng.integers(0, min(len(b), 40))):]           # cut inside the header
    elif kind == 1:
        b[int(rng.integers(0, 4))] ^= 0x20                       # no BAM magic
    else:
        b = b[:len(b) - int(rng.integers(1, 20))] if len(body_of(stream)) > 20 else b[:6]
    stream = bytes(b)
    want, err = reference_records(stream, 1 << 20)
    if err is None or want:
        pytest.skip("the stream still decodes records: the GPU test's part")
    with pytest.raises({"ValueError": ValueError, "EOFError": EOFError}[err[0]]) as e:
        list(BamParser(io.BytesIO(stream), 1 << 20))
    if "fileobj: <" in err[1]:      # the message names the file object's address
        assert "No BAM magic, instead found: " + err[1].split("found: ")[1] in str(e.value)
    else:
        assert str(e.value) == err[1]
