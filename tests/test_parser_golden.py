"""FastqParser against what the reference's parser did with the same text and buffer
size (tests/golden/parser_cases.npz, parser_errors.json; made by make_golden.py parser
from FastqParser_create_record_array, _qcmodule.c:964-1184).  The host splitter runs
everywhere; the GPU record split (sq_batch_from_fastq) is marked gpu."""
import io
import json
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN

CASES = np.load(os.path.join(GOLDEN, "parser_cases.npz"))
NAMES = [str(n) for n in CASES["names"]]
with open(os.path.join(GOLDEN, "parser_errors.json")) as f:
    ERRORS = json.load(f)
FIELDS = ("record_start", "name_length", "sequence_offset", "sequence_length", "qualities_offset",
          "tags_offset", "tags_length")


def metas_matrix(metas) -> np.ndarray:
    return np.stack([metas[f].astype(np.int64) for f in FIELDS], axis=1) if len(metas) else np.zeros((0, 7), np.int64)


def check_case(key: str, source=io.BytesIO, **parser_kwargs):
    from sequali_amd import FastqParser
    text = CASES[key.rsplit("_", 1)[0] + "_text"].tobytes()
    bs = int(CASES[key + "_buffersize"])
    arrays = list(FastqParser(source(text), bs, **parser_kwargs))
    assert [len(a) for a in arrays] == CASES[key + "_sizes"].tolist()
    assert [len(a.obj) for a in arrays] == CASES[key + "_objlens"].tolist()
    got = [metas_matrix(a._host_metas()) for a in arrays]
    got = np.concatenate(got) if got else np.zeros((0, 7), np.int64)
    assert np.array_equal(got, CASES[key + "_metas"])


def check_error(case, source=io.BytesIO, **parser_kwargs):
    from sequali_amd import FastqParser
    text = case["text"].encode("latin-1")
    parser = FastqParser(source(text), case["buffersize"], **parser_kwargs)
    if "error" not in case:
        assert [len(a) for a in parser] == case["sizes"]
        return
    exc = {"ValueError": ValueError, "EOFError": EOFError}[case["error"]]
    with pytest.raises(exc) as e:
        list(parser)
    assert str(e.value) == case["message"]


@pytest.mark.parametrize("key", NAMES)
def test_host_parser_matches_reference_chunking(key):
    check_case(key)


@pytest.mark.parametrize("i", range(len(ERRORS)))
def test_host_parser_errors_match_reference(i):
    check_error(ERRORS[i])


@pytest.mark.gpu
@pytest.mark.parametrize("key", NAMES)
def test_device_split_matches_reference_chunking(key):
    check_case(key, split_on_device=True)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(ERRORS)))
def test_device_split_errors_match_reference(i):
    check_error(ERRORS[i], split_on_device=True)


@pytest.mark.gpu
@pytest.mark.parametrize("key", NAMES)
def test_device_split_from_pinned_pages_matches_reference_chunking(key):
    """the text in page-locked memory: every buffer but the first was sent ahead by the call before
    (sq_batch_from_fastq_ahead) and is joined with its leftover inside the device"""
    from sequali_amd import PinnedReader
    check_case(key, source=PinnedReader, split_on_device=True)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(0, len(ERRORS), 3))
def test_device_split_from_pinned_pages_errors_match_reference(i):
    from sequali_amd import PinnedReader
    check_error(ERRORS[i], source=PinnedReader, split_on_device=True)


@pytest.mark.gpu
def test_upload_ahead_through_the_c_abi():
    """sq_batch_from_fastq_ahead: a buffer that contains what was sent ahead (leftover in front, more text
    behind), one that does not (the bytes sent ahead are dropped), and the tables of the records"""
    import ctypes as C
    from sequali_amd import PinnedReader, QCMetrics, synth
    from sequali_amd._lib import context, lib
    from sequali_amd._qc import FastqRecordArrayView, _DeviceBatch
    text, metas = synth.host_records(0, 0, 30_000)
    pages = PinnedReader(text)
    base, n = pages._address, len(text)
    ends = [n // 4 + 7, n // 2 + 1, 3 * n // 4, n]     # four buffers; the split leaves a leftover in front of 2, 3 and 4
    # sent ahead by call k: from in front of the next buffer (its leftover included) to the middle of it /
    # from behind the leftover to beyond the end of the next buffer / nothing
    ahead = [(base + ends[0] - 5000, 5000 + (ends[1] - ends[0]) // 2), (base + ends[1], n - ends[1]), (None, 0), (None, 0)]
    consumed = C.c_size_t(0)
    got, start = QCMetrics(), 0
    for k in range(4):
        h = lib().sq_batch_from_fastq_ahead(context(), base + start, ends[k] - start, C.byref(consumed), *ahead[k])
        assert h
        got.add_record_array(FastqRecordArrayView._from_device(_DeviceBatch(h)))
        start += consumed.value
    assert start == n
    # sent ahead but never asked for: the next call takes its whole text from the host
    h = lib().sq_batch_from_fastq_ahead(context(), base, n // 4, C.byref(consumed), base + n // 2, 1000)
    assert h
    first = _DeviceBatch(h)
    h = lib().sq_batch_from_fastq(context(), base, n // 4, C.byref(consumed))
    assert _DeviceBatch(h).number_of_records == first.number_of_records
    want = QCMetrics()
    want.add_record_array(FastqRecordArrayView._from_buffer(text, metas))
    assert got.number_of_reads == 30_000
    assert got.base_count_table() == want.base_count_table()
    assert got.phred_count_table() == want.phred_count_table()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [(0, 200_000), (2, 3000)])
def test_device_split_of_synthetic_text_equals_generator_metas(kind, n):
    """The record split of BASELINE.json's synthetic FASTQ, straight through the C ABI:
    metas equal to the generator's, `consumed` stops before a truncated last record,
    and the modules see the same records (QCMetrics tables equal)."""
    import ctypes as C
    from sequali_amd import QCMetrics, synth
    from sequali_amd._lib import context, lib
    from sequali_amd._qc import META_DTYPE, FastqRecordArrayView, _DeviceBatch
    text, metas = synth.host_records(kind, 0, n)
    cut = text + text[:100]
    consumed = C.c_size_t(0)
    h = lib().sq_batch_from_fastq(context(), cut, len(cut), C.byref(consumed))
    assert h
    batch = _DeviceBatch(h)
    assert consumed.value == len(text)
    assert batch.number_of_records == n
    got = batch.download_metas()
    for f in FIELDS:
        assert np.array_equal(got[f], metas[f]), f
    assert batch.total_bases == int(metas["sequence_length"].sum())
    assert batch.max_length == int(metas["sequence_length"].max())
    arr = FastqRecordArrayView._from_device(batch)
    a, b = QCMetrics(), QCMetrics()
    a.add_record_array(arr)
    b.add_record_array(FastqRecordArrayView._from_buffer(text, metas))
    assert a.base_count_table() == b.base_count_table()
    assert a.phred_count_table() == b.phred_count_table()
    assert np.array_equal(arr.accumulated_error_rates().view(np.uint64),
                          FastqRecordArrayView._from_buffer(text, metas).accumulated_error_rates().view(np.uint64)) \
        or b.number_of_reads() == n


def test_arrays_outlive_the_staging_blocks_they_came_from(monkeypatch):
    """The reference's arrays own their buffer and stay valid however far the parser has moved on
    (_qcmodule.c:575-579).  Arrays of sq_feeder are windows of pinned staging blocks that go back to a
    pool: a block somebody still holds arrays of keeps its bytes (in pageable memory) when it does."""
    from sequali_amd import FastqParser, _qc, synth
    monkeypatch.setattr(_qc, "_STAGE_LIMIT", 1 << 20)     # 1 MiB staging blocks: 5 MB of text are several of them
    n = 15_000
    text = synth.host_records(synth.ILLUMINA, 0, n, 0)[0]
    arrays = list(FastqParser(io.BytesIO(text)))           # every array held while the parser runs to the end
    blocks = {id(a._blk) for a in arrays}
    assert len(blocks) > _qc._Feeder.KEEP + 2
    assert sum(len(a) for a in arrays) == n
    from tests.helpers import split_fastq
    buf, metas = split_fastq(text)
    first = 0
    for a in arrays:                                       # the oldest ones first: their blocks are long gone
        assert a[0].name() == buf[int(metas["record_start"][first]):][:int(metas["name_length"][first])].decode()
        last = first + len(a) - 1
        r = a[len(a) - 1]
        s0 = int(metas["record_start"][last]) + int(metas["sequence_offset"][last])
        assert r.sequence() == buf[s0:s0 + int(metas["sequence_length"][last])].decode()
        q0 = int(metas["record_start"][last]) + int(metas["qualities_offset"][last])
        assert r.qualities() == buf[q0:q0 + int(metas["sequence_length"][last])].decode()
        first += len(a)


@pytest.mark.gpu
def test_arrays_of_released_blocks_can_still_be_counted(monkeypatch):
    """... and a module that is handed such an array later counts it (the block is uploaded from the
    pageable copy)."""
    from oracle import oracle
    from sequali_amd import FastqParser, QCMetrics, _qc, synth
    from tests.helpers import split_fastq
    monkeypatch.setattr(_qc, "_STAGE_LIMIT", 1 << 20)
    n = 15_000
    text = synth.host_records(synth.ILLUMINA, 0, n, 0)[0]
    arrays = list(FastqParser(io.BytesIO(text)))
    qc = QCMetrics()
    for a in arrays:
        qc.add_record_array(a)
    buf, metas = split_fastq(text)
    ref = oracle.QCMetrics()
    ref.add(buf, metas)
    assert qc.number_of_reads == n
    np.testing.assert_array_equal(np.array(qc.base_count_table(), np.uint64), ref.base_count_table())
    np.testing.assert_array_equal(np.array(qc.phred_scores(), np.uint64), ref.phred_scores())
    rates = np.concatenate([a.accumulated_error_rates() for a in arrays])
    np.testing.assert_array_equal(rates.view(np.uint64), metas["accumulated_error_rate"].view(np.uint64))
