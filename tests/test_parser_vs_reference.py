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


def read_outcome(parser_class, text, buffer_size, plan):
    """what a run of read(n) calls -- and next() in between -- hands out: per call (records, len(obj), the names), or the
    exception that ended it"""
    out = []
    try:
        p = parser_class(io.BytesIO(text), buffer_size)
        for n in plan:
            if n is None:
                try:
                    arr = next(p)
                except StopIteration:
                    out.append("stop")
                    continue
            else:
                arr = p.read(n)
            out.append((len(arr), len(arr.obj), [arr[i].name() for i in range(len(arr))]))
    except (ValueError, EOFError, OverflowError, TypeError) as e:
        out.append((type(e).__name__, str(e) if not isinstance(e, TypeError) else ""))
    return out


@pytest.mark.parametrize("seed", range(120))
def test_read_in_lock_step(seed):
    """FastqParser.read(number_of_records) (:1186-1245; what the driver uses to keep two files of a pair in step): any
    mixture of counts, iteration steps in between, calls at the end of the file.

    Where the reference is defined, that is.  Two of its paths are not, and are left out: (1) read(n) stops at n records
    and leaves the rest of its buffer to the next call, whose fresh buffer of `buffer_size` bytes is filled with that rest
    unchecked (:984-992) -- with a buffer smaller than the rest it writes past its allocation (seen: buffer size 1, names
    that are no ASCII); hence buffers that hold the whole text.  (2) read(n) for more records than the file still has
    parses what is there, meets the end of the file, copies the buffer to its true size -- and leaves the parsed records
    pointing into the buffer it has just released (:1040-1051); hence no count above what is left, except at the very end
    (nothing parsed, nothing dangling).  This parser hands out the records in both cases."""
    from sequali_amd import FastqParser
    rng = np.random.default_rng(15000 + seed)
    n_records = int(rng.choice([0, 1, 5, 80]))
    text = random_text(rng, n_records, int(rng.choice([1, 30, 400])))
    left, plan = n_records, []
    for _ in range(int(rng.integers(1, 12))):
        op = rng.choice([0, 1, 2, 7, 50, None, -1])
        if op is None:
            plan.append(None)
            left = 0            # the buffer holds the whole text: one step of the iteration hands out all of it
        elif op <= 0 or left == 0:
            plan.append(int(op))
        else:
            plan.append(int(min(op, left)))
            left -= plan[-1]
    for buffer_size in (len(text) + 1, 1 << 20):
        got, want = read_outcome(FastqParser, text, buffer_size, plan), read_outcome(REF.FastqParser, text, buffer_size, plan)
        assert got == want, (buffer_size, plan)
