"""FastqParser over a source the feeder reads by itself (sq_feeder_set_source_memory / _fd, csrc/sq_feed.hip: worker threads
copy the text into the staging blocks and note the line ends; the record split takes its newlines from their notes) against
the same parser fed through readinto() (the reference's way, _qcmodule.c:1040-1051): the same arrays -- sizes, bytes, every
FastqMeta field -- or the same exception, for an io.BytesIO, for a BytesIO that has been read from already, and for a file
on disk.  No GPU needed (the parser's host side)."""
import io
import os

import numpy as np
import pytest

from sequali_amd import FastqParser
from sequali_amd import _qc


class Fed(io.BytesIO):
    """a BytesIO the parser does not recognise: it is read through readinto()"""


def _text(rng, n, max_len, bad=None):
    out = []
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L).tobytes()
        qual = (rng.integers(0, 94, size=L) + 33).astype(np.uint8).tobytes()
        rec = b"@r%d some:comment\n" % i + seq + b"\n+\n" + qual + b"\n"
        if bad is not None and i == bad[0]:
            rec = bad[1](rec)
        out.append(rec)
    return b"".join(out)


def _arrays(fileobj, buffersize, read_plan=None):
    """[(n records, obj bytes, metas)] or [..., exception] of a parser over fileobj"""
    p = FastqParser(fileobj, buffersize)
    out = []
    try:
        if read_plan:
            for k in read_plan:
                a = p.read(k)
                out.append((len(a), bytes(a.obj), a._metas.tobytes()))
                if len(a) == 0:
                    break
        else:
            for a in p:
                out.append((len(a), bytes(a.obj), a._metas.tobytes()))
    except Exception as e:   # noqa: BLE001 -- compared below
        out.append((type(e), str(e)))
    return out


@pytest.mark.parametrize("seed,n,max_len,buffersize", [(1, 40_000, 300, 128 * 1024), (2, 90_000, 151, 64 * 1024), (3, 3_000, 20_000, 128 * 1024),
                                                       (4, 20_000, 200, 1000), (5, 60_000, 400, 3 << 20), (6, 5, 50, 7), (7, 0, 1, 128 * 1024)])
def test_a_bytesio_read_by_the_feeder_gives_the_arrays_of_one_read_through_readinto(seed, n, max_len, buffersize):
    rng = np.random.default_rng(seed)
    text = _text(rng, n, max_len)
    want = _arrays(Fed(text), buffersize)
    src = io.BytesIO(text)
    got = _arrays(src, buffersize)
    assert got == want
    assert src.tell() == len(text)   # left at its end, as after the reference's reads
    # the parser took the source path at all (a BytesIO of more than nothing)
    if text:
        p = FastqParser(io.BytesIO(text), buffersize)
        next(iter(p), None)
        assert p._source is not None


def test_read_n_and_a_position_in_the_middle():
    rng = np.random.default_rng(11)
    text = _text(rng, 30_000, 250)
    first = text.index(b"@r100 ")   # a record boundary
    for plan in ([1, 5, 1000, 7, 20000, 9000], [40000, 1]):
        a, b = Fed(text), io.BytesIO(text)
        a.seek(first)
        b.seek(first)
        assert _arrays(b, 128 * 1024, plan) == _arrays(a, 128 * 1024, plan)


@pytest.mark.parametrize("what", ["no_at", "no_plus", "lengths", "non_ascii", "truncated"])
def test_errors_are_those_of_the_fed_parser(what):
    """a damaged record far into the text (several pieces and a block in front of it): the arrays in front of it, then the
    same exception with the same text"""
    rng = np.random.default_rng(21)
    breaks = {"no_at": lambda r: b"X" + r[1:], "no_plus": lambda r: r.replace(b"\n+\n", b"\n-\n", 1),
              "lengths": lambda r: r[:-2] + b"\n", "non_ascii": lambda r: r[:8] + b"\xc3" + r[9:], "truncated": lambda r: r}
    text = _text(rng, 70_000, 300, bad=(61_234, breaks[what]))
    if what == "truncated":
        text = text[:-37]
    for buffersize in (128 * 1024, 5000):
        assert _arrays(io.BytesIO(text), buffersize) == _arrays(Fed(text), buffersize)


def test_a_file_on_disk_is_read_by_the_feeder(tmp_path):
    rng = np.random.default_rng(31)
    text = _text(rng, 50_000, 280)
    path = tmp_path / "reads.fastq"
    path.write_bytes(text)
    want = _arrays(Fed(text), 128 * 1024)
    with open(path, "rb") as f:
        p = FastqParser(f, 128 * 1024)
        got = [(len(a), bytes(a.obj), a._metas.tobytes()) for a in p]
        assert p._source is not None and f.tell() == os.path.getsize(path)
    assert got == want
    with open(path, "rb") as f:      # from a position that is not the file's start
        f.seek(text.index(b"@r777 "))
        g = Fed(text)
        g.seek(text.index(b"@r777 "))
        assert _arrays(f, 64 * 1024) == _arrays(g, 64 * 1024)


def test_the_switch_sends_every_file_object_through_readinto(monkeypatch):
    monkeypatch.setattr(_qc, "_USE_SOURCE", False)
    text = _text(np.random.default_rng(41), 2000, 100)
    p = FastqParser(io.BytesIO(text))
    assert sum(len(a) for a in p) == 2000 and p._source is None


def test_a_parser_dropped_in_the_middle_of_its_file():
    """The workers read from the BytesIO's bytes: those must outlive sq_feeder_free, which waits for the workers (they belong
    to the feeder object).  A parser that let go of them first crashed now and then -- here: 300 parsers dropped after one
    array of a text of several pieces."""
    rng = np.random.default_rng(51)
    text = _text(rng, 60_000, 300)
    for i in range(300):
        p = FastqParser(io.BytesIO(text), 128 * 1024)
        assert len(next(p)) > 0
        del p


# -- the walker (feed_walker, csrc/sq_feed.hip): one more thread splits the records as the text arrives and sq_feeder_next
# hands out the ones that end inside its window; the window's own record loop runs where the walker cannot answer

def test_tiny_records_fill_the_meta_area_in_front_of_the_text():
    """records of 10 bytes: a block's meta area (one meta per 96 bytes of text) is full long before its text is -- the walker
    stops there, the caller's thread makes a bigger one and goes on"""
    text = b"@a\nA\n+\nI\n" * 400_000
    for buffersize in (128 * 1024, 1 << 20):
        got = _arrays(io.BytesIO(text), buffersize)
        assert got == _arrays(Fed(text), buffersize)
        assert sum(n for n, _, _ in got) == 400_000


@pytest.mark.parametrize("what", ["no_at", "no_plus", "lengths", "non_ascii"])
def test_a_damaged_record_at_every_place_of_a_window(what):
    """the damaged record slides through the windows of a small buffer (its start, its `+`, its end on either side of a
    window's end), in the first window and far behind it"""
    rng = np.random.default_rng(61)
    breaks = {"no_at": lambda r: b"X" + r[1:], "no_plus": lambda r: r.replace(b"\n+\n", b"\n-\n", 1),
              "lengths": lambda r: r[:-2] + b"\n", "non_ascii": lambda r: r[:-3] + b"\xc3" + r[-2:]}
    for bad in list(range(0, 12)) + list(range(3000, 3040, 3)):
        text = _text(rng, 3100, 60, bad=(bad, breaks[what]))
        for buffersize in (700, 4096):
            assert _arrays(io.BytesIO(text), buffersize) == _arrays(Fed(text), buffersize), (bad, buffersize)


def test_read_n_in_the_middle_of_an_iteration():
    """FastqParser.read(n) asks for record counts the walker's windows do not know: the caller's thread takes over, in the
    middle of a block, and the arrays stay those of the fed parser"""
    rng = np.random.default_rng(71)
    text = _text(rng, 40_000, 200)

    def mixed(fileobj):
        p = FastqParser(fileobj, 64 * 1024)
        out = []
        for _ in range(20):
            a = next(p)
            out.append((len(a), bytes(a.obj), a._metas.tobytes()))
        for k in (3, 1000, 1):
            a = p.read(k)
            out.append((len(a), bytes(a.obj), a._metas.tobytes()))
        out.extend((len(a), bytes(a.obj), a._metas.tobytes()) for a in p)
        return out

    assert mixed(io.BytesIO(text)) == mixed(Fed(text))


@pytest.mark.parametrize("walker", ["0", "1", "plain"])
def test_with_and_without_the_walker(monkeypatch, walker):
    monkeypatch.setenv("SQ_FEED_WALKER", walker)
    rng = np.random.default_rng(81)
    text = _text(rng, 50_000, 300)
    assert _arrays(io.BytesIO(text), 128 * 1024) == _arrays(Fed(text), 128 * 1024)
    text = _text(rng, 20_000, 300, bad=(15_000, lambda r: r.replace(b"\n+\n", b"\n-\n", 1)))
    assert _arrays(io.BytesIO(text), 128 * 1024) == _arrays(Fed(text), 128 * 1024)


def test_the_workers_plain_copy(monkeypatch):
    """SQ_FEED_COPY=plain: memcpy, then the scan over the block (the way before the fused pass) -- the same arrays"""
    monkeypatch.setenv("SQ_FEED_COPY", "plain")
    rng = np.random.default_rng(91)
    text = _text(rng, 50_000, 300)
    assert _arrays(io.BytesIO(text), 128 * 1024) == _arrays(Fed(text), 128 * 1024)
    text = _text(rng, 20_000, 300, bad=(15_000, lambda r: r[:8] + b"\xc3" + r[9:]))
    assert _arrays(io.BytesIO(text), 128 * 1024) == _arrays(Fed(text), 128 * 1024)
