"""sequali_amd.FastqParser (the host splitter of csrc/sq_feed.hip + sq_hostsimd.cpp: product code that needs no GPU)
against the reference's own FastqParser (oracle/_ref, FastqParser_create_record_array _qcmodule.c:964-1184) on random
texts, random buffer sizes and random damage: the same arrays (record counts, the length of each array's bytes, every
record's name, sequence and qualities) or the same exception with the same message.  tests/test_parser_golden.py pins
chosen cases from fixtures; this looks where nobody chose.  CPU only; skipped where oracle/_ref is absent."""
import io

import numpy as np
import pytest

from tests.test_oracle_vs_reference import REF, fastq

pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")


def random_text(rng, n, max_len):
    names, seqs, quals = [], [], []
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        names.append("".join(chr(c) for c in rng.integers(33, 127, size=int(rng.integers(0, 40)))) + (" tag=1" if i % 3 == 0 else ""))
        seqs.append(rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), size=L).tobytes().decode())
        quals.append((rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes().decode())   # '@' and '+' at a line's start included
    return fastq(names, seqs, quals)


def damage(rng, text: bytes) -> bytes:
    b = bytearray(text)
    for _ in range(int(rng.integers(1, 4))):
        if not b:
            break
        at = int(rng.integers(0, len(b)))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            del b[at]
        elif kind == 1:
            b.insert(at, int(rng.choice([10, 64, 43, 65, 13, 0, 200])))
        elif kind == 2:
            b[at] = int(rng.choice([10, 64, 43, 13, 32, 0, 255]))
        elif kind == 3:
            del b[at:]                          # a truncated file
        else:
            b[at:at] = b"\n"
    return bytes(b)


def outcome(parser_class, text, buffer_size):
    """([(records, len(obj), [(name, sequence, qualities)])], None) or what was read before the error + (type, message)"""
    arrays = []
    try:
        for arr in parser_class(io.BytesIO(text), buffer_size):
            recs = [(arr[i].name(), arr[i].sequence(), arr[i].qualities()) for i in range(len(arr))]
            arrays.append((len(arr), len(arr.obj), recs))
    except (ValueError, EOFError, OverflowError) as e:
        return arrays, (type(e).__name__, str(e))
    return arrays, None


@pytest.mark.parametrize("seed", range(60))
def test_texts_nobody_chose(seed):
    from sequali_amd import FastqParser
    rng = np.random.default_rng(13000 + seed)
    text = random_text(rng, int(rng.choice([0, 1, 3, 40, 400])), int(rng.choice([0, 5, 60, 300, 3000])))
    for buffer_size in (1, 7, 64, 1000, 4096, 1 << 20):
        assert outcome(FastqParser, text, buffer_size) == outcome(REF.FastqParser, text, buffer_size), buffer_size


@pytest.mark.parametrize("seed", range(400))
def test_damaged_texts(seed):
    from sequali_amd import FastqParser
    rng = np.random.default_rng(14000 + seed)
    text = damage(rng, random_text(rng, int(rng.choice([1, 2, 5, 60])), int(rng.choice([1, 8, 70, 500]))))
    for buffer_size in (1, 33, 512, 1 << 16):
        got, want = outcome(FastqParser, text, buffer_size), outcome(REF.FastqParser, text, buffer_size)
        assert got == want, (buffer_size, text[:200])
