"""Host-side mirror of the reference's ``sequali._qc`` extension for the QC hot
path (src/sequali/_qc.pyi:45-188): same class names, constructor arguments,
method names, return types and error behaviour, with the accumulation done by
the HIP kernels behind libsqgpu.so (include/sqgpu.h).

Deviations that follow from deferred execution on a stream:
  * ``add_record_array`` enqueues work and returns; an invalid phred byte is
    reported as ``ValueError`` by the next getter / ``flush()`` (``add_read`` is
    synchronous and raises at once, like the reference);
  * ``accumulated_error_rate`` is written into the array's metas when the
    QCMetrics object is flushed (any getter flushes).
"""
from __future__ import annotations

import array
import ctypes as C
import io
import sys
import os
import warnings
import weakref
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check, context, lib

# constants exported by the reference module (_qcmodule.c:6082-6171)
NUMBER_OF_NUCS = 5
NUMBER_OF_PHREDS = 12
TABLE_SIZE = NUMBER_OF_PHREDS * NUMBER_OF_NUCS
PHRED_MAX = 93
A, C_, G, T, N = 0, 1, 2, 3, 4
MAX_SEQUENCE_SIZE = 64
DEFAULT_END_ANCHOR_LENGTH = 100
DEFAULT_MAX_UNIQUE_FRAGMENTS = 5_000_000
DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS = 1_000_000
DEFAULT_FRAGMENT_LENGTH = 21
DEFAULT_UNIQUE_SAMPLE_EVERY = 8
DEFAULT_BASES_FROM_START = 100
DEFAULT_BASES_FROM_END = 100
DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH = 8
DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH = 8
DEFAULT_FINGERPRINT_FRONT_SEQUENCE_OFFSET = 64
DEFAULT_FINGERPRINT_BACK_SEQUENCE_OFFSET = 64
INSERT_SIZE_MAX_ADAPTER_STORE_SIZE = 31

META_DTYPE = np.dtype([
    ("record_start", "<u8"), ("name_length", "<u4"), ("sequence_offset", "<u4"),
    ("sequence_length", "<u4"), ("qualities_offset", "<u4"), ("tags_offset", "<u4"),
    ("tags_length", "<u4"), ("accumulated_error_rate", "<f8")])
assert META_DTYPE.itemsize == 40

# SCORE_TO_ERROR_RATE (score_to_error_rate.h), same recipe as its generator script
_ERROR_RATES = np.array([10 ** -(q / 10) for q in range(PHRED_MAX + 1)], dtype=np.float64)


def _addr(b) -> int:
    if isinstance(b, np.ndarray):
        return b.ctypes.data
    if len(b) == 0:
        return 0
    return np.frombuffer(b, dtype=np.uint8).ctypes.data


def _type_name(obj) -> str:
    return repr(type(obj))


# ---------------------------------------------------------------------------
# record boundary
# ---------------------------------------------------------------------------
class FastqRecordView:
    """FastqRecordView__new__, _qcmodule.c:372-482"""

    __slots__ = ("obj", "_meta")

    def __init__(self, name: str, sequence: str, qualities: str,
                 tags: Optional[bytes] = None):
        for label, v in (("name", name), ("sequence", sequence), ("qualities", qualities)):
            if not isinstance(v, str):
                raise TypeError(f"FastqRecordView() argument '{label}' must be str, "
                                f"not {type(v).__name__}")
        if tags is not None and not isinstance(tags, bytes):
            raise TypeError(f"FastqRecordView() argument 'tags' must be bytes, "
                            f"not {type(tags).__name__}")
        if not name.isascii():
            raise ValueError(f"name should contain only ASCII characters: {name!r}")
        if not sequence.isascii():
            raise ValueError(f"sequence should contain only ASCII characters: {sequence!r}")
        if not qualities.isascii():
            raise ValueError(f"qualities should contain only ASCII characters: {sequence!r}")
        if len(sequence) != len(qualities):
            raise ValueError("sequence and qualities have different lengths: "
                             f"{len(sequence)} and {len(qualities)}")
        tags = tags or b""
        total = len(name) + 2 * len(sequence) + len(tags)
        if total > 0xFFFFFFFF:
            raise OverflowError("Total length of FASTQ record exceeds 4 GiB. "
                                f"Record name: {name!r}")
        qb = qualities.encode("ascii")
        q = np.frombuffer(qb, dtype=np.uint8).astype(np.int16) - 33
        bad = np.nonzero((q < 0) | (q > PHRED_MAX))[0]
        if len(bad):
            raise ValueError(f"Not a valid phred character: {qualities[int(bad[0])]}")
        # sequential sum, as the loop at :443-451 (cumsum adds left to right)
        err = float(np.cumsum(_ERROR_RATES[q])[-1]) if len(q) else 0.0
        self.obj = name.encode("ascii") + sequence.encode("ascii") + qb + tags
        m = np.zeros(1, dtype=META_DTYPE)
        m["name_length"] = len(name)
        m["sequence_offset"] = len(name)
        m["sequence_length"] = len(sequence)
        m["qualities_offset"] = len(name) + len(sequence)
        m["tags_offset"] = len(name) + 2 * len(sequence)
        m["tags_length"] = len(tags)
        m["accumulated_error_rate"] = err
        self._meta = m

    @classmethod
    def _from(cls, obj: bytes, meta: np.ndarray) -> "FastqRecordView":
        self = cls.__new__(cls)
        self.obj = obj
        self._meta = meta
        return self

    def _slice(self, off_field: str, length: int) -> bytes:
        m = self._meta[0]
        start = int(m["record_start"]) + (int(m[off_field]) if off_field else 0)
        return self.obj[start:start + length]

    def name(self) -> str:
        return self._slice("", int(self._meta[0]["name_length"])).decode("ascii")

    def sequence(self) -> str:
        return self._slice("sequence_offset", int(self._meta[0]["sequence_length"])).decode("ascii")

    def qualities(self) -> str:
        return self._slice("qualities_offset", int(self._meta[0]["sequence_length"])).decode("ascii")

    def tags(self) -> bytes:
        return self._slice("tags_offset", int(self._meta[0]["tags_length"]))


class _DeviceBatch:
    """A record array resident in HBM (sq_batch)."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                lib().sq_batch_free(self.handle)
        except Exception:
            pass
        self.handle = None

    @property
    def number_of_records(self) -> int:
        return lib().sq_batch_size(self.handle)

    @property
    def total_bases(self) -> int:
        return lib().sq_batch_total_bases(self.handle)

    @property
    def max_length(self) -> int:
        return lib().sq_batch_max_length(self.handle)

    def download(self):
        """(bytes, metas) copied back from HBM"""
        nbytes, n = lib().sq_batch_bytes(self.handle), self.number_of_records
        buf = np.zeros(nbytes, dtype=np.uint8)
        metas = np.zeros(n, dtype=META_DTYPE)
        check(lib().sq_batch_download(self.handle, buf.ctypes.data, nbytes, metas.ctypes.data, n))
        return buf, metas

    def download_metas(self) -> np.ndarray:
        metas = np.zeros(self.number_of_records, dtype=META_DTYPE)
        check(lib().sq_batch_download(self.handle, None, 0, metas.ctypes.data, len(metas)))
        return metas

    def error_rates(self) -> np.ndarray:
        out = np.zeros(self.number_of_records, dtype=np.float64)
        if len(out):
            check(lib().sq_batch_error_rates(self.handle, out.ctypes.data, len(out)))
        return out


class FastqRecordArrayView:
    """FastqRecordArrayView, _qcmodule.c:575-883: one bytes object plus the
    40-byte metas of its records.  Built from views it copies them into one new
    buffer (and, unlike :672-684, points the metas at that new buffer)."""

    def __init__(self, view_items: Iterable[FastqRecordView]):
        try:
            items = list(view_items)
        except TypeError:
            raise TypeError("view_items should be iterable")
        metas = np.zeros(len(items), dtype=META_DTYPE)
        parts = []
        pos = 0
        for i, item in enumerate(items):
            if not isinstance(item, FastqRecordView):
                raise TypeError("Expected an iterable of FastqRecordView objects, but item "
                                f"{i} is of type {type(item)!r}: {item!r}")
            m = item._meta[0]
            nl, sl, tl = int(m["name_length"]), int(m["sequence_length"]), int(m["tags_length"])
            start = int(m["record_start"])
            parts.append(item.obj[start:start + nl])
            so, qo, to = (int(m["sequence_offset"]), int(m["qualities_offset"]), int(m["tags_offset"]))
            parts.append(item.obj[start + so:start + so + sl])
            parts.append(item.obj[start + qo:start + qo + sl])
            parts.append(item.obj[start + to:start + to + tl])
            metas[i] = (pos, nl, nl, sl, nl + sl, nl + 2 * sl, tl, m["accumulated_error_rate"])
            pos += nl + 2 * sl + tl
        self.obj = b"".join(parts)
        self._metas = metas
        self._batch: Optional[_DeviceBatch] = None
        self._writeback = None  # weakref to the QCMetrics that owes this array its error rates

    @classmethod
    def _from_buffer(cls, obj, metas: np.ndarray) -> "FastqRecordArrayView":
        self = cls.__new__(cls)
        self.obj = obj
        self._metas = metas
        self._batch = None
        self._writeback = None
        return self

    @classmethod
    def _from_device(cls, batch: _DeviceBatch) -> "FastqRecordArrayView":
        """An array whose records only live in HBM (synthetic data, GPU-side parsers)."""
        self = cls.__new__(cls)
        self.obj = None
        self._metas = None
        self._batch = batch
        self._writeback = None
        return self

    def __len__(self) -> int:
        if self._metas is None:
            return self._batch.number_of_records
        return len(self._metas)

    def __getitem__(self, i: int) -> FastqRecordView:
        n = len(self)
        if i < 0:
            i += n      # what the interpreter does before it calls a C type's sq_item ...
            if i < 0:
                i += n  # ... and what _qcmodule.c:693-695 then does once more: arr[-len - 1] is the LAST record there, and here
        if i < 0 or i >= n:
            raise IndexError("array index out of range")
        metas = self._host_metas()
        return FastqRecordView._from(self.obj, metas[i:i + 1])

    def _host_metas(self) -> np.ndarray:
        """the metas (and, for an array that was built in HBM, its bytes) on the host:
        fetched once, when a record is first looked at"""
        if self._metas is None:
            if self.obj is None:
                buf, self._metas = self._batch.download()
                self.obj = buf.tobytes()
            else:
                self._metas = self._batch.download_metas()  # split on the device
        return self._metas

    def is_mate(self, other) -> bool:
        """FastqRecordArrayView_is_mate, _qcmodule.c:814-850"""
        if not isinstance(other, FastqRecordArrayView):
            raise TypeError(f"other must be of type FastqRecordArrayView, got {type(other)!r}")
        if len(self) != len(other):
            raise ValueError("other is not the same length as this record array view. "
                             f"This length: {len(self)}, other length: {len(other)}")
        m1, m2 = self._host_metas(), other._host_metas()
        return bool(lib().sq_names_are_mates(_addr(self.obj), m1.ctypes.data, _addr(other.obj), m2.ctypes.data,
                                             len(self)))

    # -- device side ---------------------------------------------------------
    def _device(self) -> _DeviceBatch:
        if self._batch is None:
            st = getattr(self, "_staged", None)
            if st is not None:
                # The array lies in a staging block already (a module that defers its work copied it there): its records in
                # HBM are the block's, not a second upload.  With two copies a QCMetrics pass that was staged wrote
                # accumulated_error_rate (:2126) into one and a NanoStats that ran on the array itself read the other
                # (scripts/fuzz.py 200 55, iteration 4: a pair whose mate was too large to stage sent InsertSizeMetrics --
                # and with it this array -- down the unstaged path between the two).
                blk, slot = st
                v = blk.view(slot, slot + 1)
                self._staged_view = v          # keeps the view (and through it the block's batch) alive
                self._batch = v._device()
                return self._batch
            h = lib().sq_batch_upload(context(), _addr(self.obj), len(self.obj),
                                      self._metas.ctypes.data, len(self._metas))
            if not h:
                raise MemoryError(_lib.last_error())
            self._batch = _DeviceBatch(h)
        return self._batch

    def _release_device(self) -> None:
        if self._metas is not None:
            self._batch = None

    def accumulated_error_rates(self) -> np.ndarray:
        """FastqMeta.accumulated_error_rate of every record (what NanoStats reads,
        _qcmodule.c:5314); flushes the QCMetrics pass that computes it."""
        if self._writeback is not None:
            qcm = self._writeback()
            if qcm is not None:
                qcm.flush()
        if self._metas is None:
            return self._batch.error_rates()
        return self._metas["accumulated_error_rate"].copy()


def _require_array(obj, what="record_array") -> FastqRecordArrayView:
    if not isinstance(obj, FastqRecordArrayView):
        raise TypeError(f"{what} should be a FastqRecordArrayView object, got {type(obj)!r}")
    return obj


def _require_view(obj) -> FastqRecordView:
    if not isinstance(obj, FastqRecordView):
        raise TypeError(f"read should be a FastqRecordView object, got {type(obj)!r}")
    return obj


def _array_of_sequences(seqs: List[str]) -> FastqRecordArrayView:
    """records that only carry a sequence (add_sequence / add_sequence_pair)"""
    metas = np.zeros(len(seqs), dtype=META_DTYPE)
    parts, pos = [], 0
    for i, s in enumerate(seqs):
        b = s.encode("ascii")
        parts.append(b)
        metas[i] = (pos, 0, 0, len(b), 0, len(b), 0, 0.0)
        pos += len(b)
    return FastqRecordArrayView._from_buffer(b"".join(parts), metas)



# ---------------------------------------------------------------------------------------------
# Staging of small arrays (the reference hands over ~128 KiB of text per call, _qcmodule.c:915;
# __main__.py:279-306 calls every module once per array).  A kernel launch per 380 reads is
# launch bound, so arrays that only live on the host are copied (the caller's array is borrowed
# for the call only, :575-607) into a staging block per source, and a module remembers which
# stretch of which block it still has to count.  A block is uploaded once, by whoever needs it
# first, when it holds _STAGE_LIMIT bytes, or when a getter, a flush or anything else that looks
# at a module's state (its `_h`) asks for it: O(1) launches per 128 MiB and module whatever the
# size of the arrays.  Deferred work surfaces errors and warnings when it runs, not in the call
# that brought the array (already so for QCMetrics' invalid phred character).
_STAGE_LIMIT = int(os.environ.get("SQ_STAGE_BYTES", str(128 << 20)))   # 0: no staging (64 MiB until round 6: what a block costs is per block -- the passes' read-backs, the move to the next block -- 28 -> 22 ms per 2 M reads with six modules, profiles/r6/exp_stage_bytes.txt)
_STAGE_ARRAY_MAX = 8 << 20      # arrays from this size on are uploaded on their own
staging_stats = {"blocks": 0, "runs": 0}   # blocks uploaded, module launches on (parts of) blocks
_USE_FEEDER = os.environ.get("SQ_FEEDER", "1") != "0"   # FastqParser over pinned staging blocks (csrc/sq_feed.hip)


class _Source:
    """what arrays that are staged together have in common (a parser); its open block dies with it"""
    __slots__ = ("open_block", "__weakref__")

    def __init__(self):
        self.open_block = None


_DEFAULT_SOURCE = _Source()      # arrays the caller built himself


class _Block:
    """arrays of one source, back to back, as one record array"""

    def __init__(self, source):
        self.source = source
        self.parts: List[bytes] = []
        self.metas: List[np.ndarray] = []
        self.arrays: List = []       # weak references: a block must not keep the caller's arrays alive (nor they each other)
        self.starts = [0]            # record index of every array, and the total behind the last
        self.nbytes = 0
        self.array: Optional["FastqRecordArrayView"] = None   # set by seal()
        self.writers: List = []      # weakrefs of QCMetrics objects that still owe it error rates

    @property
    def sealed(self) -> bool:
        return self.array is not None

    @property
    def n_arrays(self) -> int:
        return len(self.starts) - 1

    def append(self, arr: "FastqRecordArrayView") -> int:
        m = arr._metas.copy()
        m["record_start"] += self.nbytes
        self.parts.append(bytes(arr.obj))
        self.metas.append(m)
        self.arrays.append(weakref.ref(arr))
        self.starts.append(self.starts[-1] + len(m))
        self.nbytes += len(arr.obj)
        return self.n_arrays - 1

    def seal(self) -> None:
        if self.array is None:
            metas = np.concatenate(self.metas) if self.metas else np.zeros(0, dtype=META_DTYPE)
            self.array = FastqRecordArrayView._from_buffer(b"".join(self.parts), metas)
            staging_stats["blocks"] += 1
            self.parts, self.metas = [], []
            if self.source.open_block is self:
                self.source.open_block = None

    def _children_of(self, s0: int, s1: int):
        """(weak reference to the array or None, first record, behind its last record) relative to array s0"""
        return [(self.arrays[k] if k < len(self.arrays) else None, self.starts[k] - self.starts[s0],
                 self.starts[k + 1] - self.starts[s0]) for k in range(s0, s1)]

    def view(self, s0: int, s1: int) -> "FastqRecordArrayView":
        """the arrays [s0, s1) of the block as one record array in HBM (the block itself when
        that is all of them: the common case, every module sees every array)"""
        self.seal()
        if s0 == 0 and s1 == self.n_arrays:
            v = self.array
        else:
            parent = self.array._device()
            r0, r1 = self.starts[s0], self.starts[s1]
            h = lib().sq_batch_view(parent.handle, r0, r1 - r0)
            if not h:
                raise MemoryError(_lib.last_error())
            v = FastqRecordArrayView._from_device(_DeviceBatch(h))
            v.obj = self.array.obj
            v._metas = self.array._metas[r0:r1] if self.array._metas is not None else None
            v._parent = parent          # the memory belongs to the block's batch
            v._blk, v._blk_r0 = getattr(self.array, "_blk", None), r0    # (a weak reference: see seal)
        # (one list per stretch: the block is sealed, and every module of a driver loop asks for the same stretch)
        cache = self.__dict__.setdefault("_children_lists", {})
        children = cache.get((s0, s1))
        if children is None:
            children = cache[(s0, s1)] = self._children_of(s0, s1)
        v._children = children
        return v


_LIVE_FEEDERS = weakref.WeakSet()   # at exit their workers are stopped while what they read from is still there


def _stop_feeders() -> None:
    for f in list(_LIVE_FEEDERS):
        f.__del__()


import atexit  # noqa: E402
atexit.register(_stop_feeders)


class _Feeder:
    """sq_feeder (csrc/sq_feed.hip): FastqParser's buffer logic over pinned staging blocks"""

    KEEP = 2     # sealed blocks whose pinned host copy is kept (an array the caller still holds may be asked for its bytes)

    def __init__(self, read_in_size: int):
        try:
            ctx = context()
        except Exception:      # no device: the host parser still works (plain memory, no upload)
            ctx = None
        self.h = lib().sq_feeder_new(ctx, read_in_size, _STAGE_LIMIT)
        if not self.h:
            raise MemoryError(_lib.last_error())
        self.source = None     # what the feeder's workers read from (kept alive until they have stopped: __del__)
        self.sealed: List[int] = []
        self.expects = False   # sq_feeder_expect_uploads has been called
        _LIVE_FEEDERS.add(self)

    def __del__(self):
        try:
            if self.h:
                lib().sq_feeder_free(self.h)     # waits for the workers
        except Exception:
            pass
        self.h = None
        self.source = None

    def expect_uploads(self) -> None:
        self.expects = True
        if self.h:
            lib().sq_feeder_expect_uploads(self.h)

    def retire(self, blk: "_FeedBlock") -> None:
        """a block has been closed: the oldest pinned copies go back to the pool.  A block that somebody
        still holds arrays of and that has no copy in HBM keeps its bytes: they move to pageable memory
        (the reference's arrays own their buffer and stay valid for ever, _qcmodule.c:575-579)"""
        self.sealed.append((blk.block_id, weakref.ref(blk)))
        while len(self.sealed) > self.KEEP:
            block_id, ref = self.sealed.pop(0)
            old = ref()
            if old is not None and old.array is None:
                old.detach()
            lib().sq_feeder_release(self.h, block_id)


_MAX_RECORDS = (1 << 63) - 1
_USE_SOURCE = os.environ.get("SQ_FEEDER_SOURCE", "1") != "0"   # 0: every file object is read through its readinto()


def _feeder_source(f: "_Feeder", fileobj):
    """Lets the feeder read `fileobj` by itself where that is the same bytes (sq_feeder_set_source_*: worker threads copy
    the text into the staging blocks and note the line ends on their way -- neither readinto() nor the newline search run
    on the thread of the caller's loop): an io.BytesIO (its buffer, from its position on) and a plain buffered binary
    file (its descriptor, by pread).  Returns what must stay alive while the feeder reads, or None: the file object is
    read through readinto() as the reference does (_qcmodule.c:1040-1051).  The file object's own position is left at
    its end either way once the parser has seen everything."""
    if not _USE_SOURCE or f.h is None:
        return None
    try:
        if type(fileobj) is io.BytesIO:
            pos = fileobj.tell()
            data = fileobj.getvalue()             # the bytes object it was made from, not a copy (getbuffer() un-shares it: a copy of everything)
            n = len(data) - pos
            if n <= 0:
                return None
            addr = C.cast(C.c_char_p(data), C.c_void_p).value
            if lib().sq_feeder_set_source_memory(f.h, addr + pos, n) != 0:
                return None
            fileobj.seek(0, 2)
            return (fileobj, data)                # `data` stays alive and is immutable: the workers read from it
        if type(fileobj) is io.BufferedReader and type(getattr(fileobj, "raw", None)) is io.FileIO:
            st = os.fstat(fileobj.fileno())
            import stat as _stat
            if not _stat.S_ISREG(st.st_mode):
                return None
            pos = fileobj.tell()
            n = st.st_size - pos
            if n <= 0 or lib().sq_feeder_set_source_fd(f.h, fileobj.fileno(), pos, n) != 0:
                return None
            fileobj.seek(0, 2)
            return (fileobj,)
    except (OSError, ValueError, BufferError, TypeError):
        pass
    return None


class _FeedArrayInfo(C.Structure):
    _fields_ = [("block_id", C.c_uint64), ("byte_start", C.c_uint64), ("byte_len", C.c_uint64),
                ("first_record", C.c_uint64), ("n_records", C.c_uint64)]


class _FeedBlock(_Block):
    """a pinned staging block of a FastqParser: its arrays are windows of it, its metas were
    written once, relative to the block (what the device wants)"""

    def __init__(self, source, feeder: _Feeder, block_id: int):
        _Block.__init__(self, source)
        self.feeder, self.block_id = feeder, block_id
        self.closed = False          # the parser has moved on to another block
        self.rates_version = 0       # bumped when a QCMetrics pass has left error rates in the block's metas in HBM
        self._host_text = None
        self._host_metas = None
        self.detached = False        # the pinned copy is gone and no copy in HBM was made: _host_text / _host_metas hold the block

    def detach(self) -> None:
        """the block's bytes and metas out of the pinned block (which is about to be handed back)"""
        f = self.feeder
        p = lib().sq_feeder_block_text(f.h, self.block_id)
        pm = lib().sq_feeder_block_metas(f.h, self.block_id)
        if not p or not pm:
            return
        nbytes = lib().sq_feeder_block_bytes(f.h, self.block_id)
        nrec = lib().sq_feeder_block_records(f.h, self.block_id)
        self._host_text = C.string_at(p, nbytes)
        self._host_metas = (np.ctypeslib.as_array((C.c_uint8 * (40 * nrec)).from_address(pm)).view(META_DTYPE).copy()
                            if nrec else np.zeros(0, dtype=META_DTYPE))
        self.detached = True

    @property
    def sealed(self) -> bool:
        return self.closed or self.array is not None

    def add(self, n_records: int) -> int:
        self.starts.append(self.starts[-1] + n_records)
        return self.n_arrays - 1

    def seal(self) -> None:
        if self.array is None:
            f = self.feeder
            if not self.closed:
                check(lib().sq_feeder_seal(f.h))     # the parser's next array opens a new block
                self.closed = True
                f.retire(self)
            if self.detached:    # the pinned block went back before anybody asked for a copy in HBM
                self.array = FastqRecordArrayView._from_buffer(self._host_text, self._host_metas)
            else:
                h = lib().sq_feeder_upload(f.h, self.block_id)
                if not h:
                    raise MemoryError(_lib.last_error())
                self.array = FastqRecordArrayView._from_device(_DeviceBatch(h))
            # weak: block -> array -> block would keep both (and the feeder's page-locked blocks) until the
            # cycle collector comes by, long after the parser is gone
            self.array._blk = weakref.ref(self)
            staging_stats["blocks"] += 1
            if self.source.open_block is self:
                self.source.open_block = None

    def _children_of(self, s0: int, s1: int):
        return [(None, self.starts[k] - self.starts[s0], self.starts[k + 1] - self.starts[s0]) for k in range(s0, s1)]

    # -- what an array of the block asks for when the caller looks at its records ------------
    def text(self, b0: int, n: int) -> bytes:
        p = lib().sq_feeder_block_text(self.feeder.h, self.block_id) if self._host_text is None else None
        if p:
            return C.string_at(p + b0, n)
        if self._host_text is None:     # the pinned copy is gone: from HBM
            self._host_text, self._host_metas = self.array._batch.download()
            self._host_text = self._host_text.tobytes()
        return self._host_text[b0:b0 + n]

    def host_metas(self, r0: int, r1: int) -> np.ndarray:
        """the metas of records [r0, r1), record_start relative to the block (a copy)"""
        if self._host_metas is None:
            p = lib().sq_feeder_block_metas(self.feeder.h, self.block_id)
            if p:
                return np.ctypeslib.as_array((C.c_uint8 * (40 * (r1 - r0))).from_address(p + 40 * r0)).view(META_DTYPE).copy()
            self._host_metas = self.array._batch.download_metas()
        return self._host_metas[r0:r1].copy()

    def error_rates(self, r0: int, r1: int) -> Optional[np.ndarray]:
        if self.array is None or self.array._batch is None:
            return None
        return self.array._batch.error_rates()[r0:r1]


class _FedArray(FastqRecordArrayView):
    """a record array of a FastqParser: a window of a pinned staging block.  `obj` and the metas
    are made when somebody looks at them; the modules only note which block and slot it is."""

    # (what every new array starts with lives on the class: the parser makes 5000 of these per 2 M reads)
    _batch = None
    _writeback = None
    _obj = None
    _m = None
    _m_version = -1

    def __init__(self, blk: _FeedBlock, slot: int, b0: int, blen: int, r0: int, n: int):
        self._blk, self._slot, self._b0, self._blen, self._r0, self._n = blk, slot, b0, blen, r0, n
        self._staged = (blk, slot)

    def __len__(self) -> int:
        return self._n

    @property
    def obj(self):
        if self._obj is None:
            self._obj = self._blk.text(self._b0, self._blen)
        return self._obj

    @obj.setter
    def obj(self, v):
        self._obj = v

    @property
    def _metas(self):
        if self._m is None:
            m = self._blk.host_metas(self._r0, self._r0 + self._n)
            m["record_start"] -= self._b0
            self._m = m
        if self._m_version != self._blk.rates_version:      # a QCMetrics pass has been through (:2126)
            rates = self._blk.error_rates(self._r0, self._r0 + self._n)
            if rates is not None:
                self._m["accumulated_error_rate"] = rates
            self._m_version = self._blk.rates_version
        return self._m

    @_metas.setter
    def _metas(self, v):
        self._m = v

    def accumulated_error_rates(self) -> np.ndarray:
        """FastqMeta.accumulated_error_rate of the array's records (:2126): the QCMetrics objects that
        were handed arrays of the block count what they owe first"""
        for w in list(self._blk.writers):
            q = w()
            if q is not None:
                q.flush()
        return self._metas["accumulated_error_rate"].copy()


def _stage(arr: "FastqRecordArrayView"):
    """(block, slot) of an array; an array is staged once, whoever sees it first"""
    st = getattr(arr, "_staged", None)
    if st is None:
        source = getattr(arr, "_source", None) or _DEFAULT_SOURCE
        blk = source.open_block
        if blk is None:
            blk = source.open_block = _Block(source)
        st = arr._staged = (blk, blk.append(arr))
        if blk.nbytes >= _STAGE_LIMIT:
            blk.seal()
    return st


class _Deferring:
    """what a module needs to count staged arrays later: `_todo` holds [block, first array, behind
    the last array] (or two of those for pairs), in call order; `_h` -- the C handle everything
    else goes through -- runs it first, so nothing can look at the module's state in front of
    work it still owes."""

    def _init_defer(self, handle) -> None:
        self._handle = handle
        self._todo: List[list] = []
        self._upstream: List = []       # FusedPass objects that feed this module (they may owe it work
                                        # when the caller has let go of them)

    @property
    def _h(self):
        self._drain()
        return self._handle

    @staticmethod
    def _small(arr) -> bool:
        if type(arr) is _FedArray:
            return True
        return (_STAGE_LIMIT > 0 and arr._batch is None and arr._metas is not None and
                len(arr.obj) < _STAGE_ARRAY_MAX and getattr(arr, "_parent", None) is None)

    def _enqueue(self, arr) -> bool:
        """True: the array is staged and will be counted later"""
        t = self._todo
        if type(arr) is _FedArray:      # the driver loop's case: the next array of the parser's open block
            blk, slot = arr._staged
            if t:
                last = t[-1]
                if last[0] is blk and last[2] == slot and len(last) == 3:
                    last[2] = slot + 1   # the block's writers are known, nothing has been sealed since the last call
                    return True
            if not blk.feeder.expects:   # a module takes the parser's arrays: its blocks go to HBM while they fill
                blk.feeder.expect_uploads()
        elif not self._small(arr):
            self._drain()
            return False
        else:
            blk, slot = _stage(arr)
        if t and len(t[-1]) == 3 and t[-1][0] is blk and t[-1][2] == slot:
            t[-1][2] = slot + 1
        else:
            t.append([blk, slot, slot + 1])
        self._staged_in(blk, arr)
        self._drain(sealed_only=True)
        return True

    def _enqueue_pair(self, a1, a2) -> bool:
        if not (self._small(a1) and self._small(a2)):
            self._drain()
            return False
        (b1, s1), (b2, s2) = _stage(a1), _stage(a2)
        for b in (b1, b2):
            f = getattr(b, "feeder", None)
            if f is not None and not f.expects:   # the parsers' blocks go to HBM while they fill
                f.expect_uploads()
        t = self._todo
        if t and len(t[-1]) == 6 and t[-1][0] is b1 and t[-1][2] == s1 and t[-1][3] is b2 and t[-1][5] == s2:
            t[-1][2], t[-1][5] = s1 + 1, s2 + 1
        else:
            t.append([b1, s1, s1 + 1, b2, s2, s2 + 1])
        self._drain(sealed_only=True)
        return True

    def _staged_in(self, blk, arr) -> None:
        pass

    def flush(self) -> None:
        """counts what is still staged (and raises / warns what that work raises / warns)"""
        self._drain()

    def _add_now(self, arr) -> None:
        """the single-item entry points (add_read, add_sequence): counted inside the call, with its
        warnings and exceptions, as the reference's tests use them"""
        self._drain()
        self._run(arr)

    def _add_pair_now(self, a1, a2) -> None:
        self._drain()
        self._run_pair(a1, a2)

    def _drain(self, sealed_only: bool = False, through=None) -> None:
        """runs the owed work in order; sealed_only: only blocks that are full already;
        through: stop once nothing of that block is left"""
        for f in self._upstream:
            f._drain_own(sealed_only, through)
        self._drain_own(sealed_only, through)

    def _drain_own(self, sealed_only: bool = False, through=None) -> None:
        # Not re-entrant: running an entry first lets the QCMetrics objects that write into the block go ahead
        # (_before_run), and a QCMetrics that THIS pass feeds sends the call straight back here -- the nested call then
        # ran the entries behind the one being run first, i.e. the batches of a block in REVERSE order.  Sums do not care;
        # PerTileQuality (it stops for good at the first header that does not parse, :3137-3148) and InsertSizeMetrics
        # (the first max_adapters remainders in pair order, :5570-5611) do (scripts/fuzz.py 200 11, iteration 192).  The
        # drain in progress gets to everything that is owed, in order.
        if getattr(self, "_draining", False):
            if through is not None:      # the caller wants this block's work done: the drain in progress sees to it
                self._owed_through = through       # before it returns (it may itself be limited to full blocks or to another block)
            return
        self._draining = True
        try:
            self._drain_entries(sealed_only, through)
            while getattr(self, "_owed_through", None) is not None:
                owed, self._owed_through = self._owed_through, None
                self._drain_entries(False, owed)
        finally:
            self._draining = False
            self._owed_through = None

    def _drain_entries(self, sealed_only: bool, through) -> None:
        while self._todo:
            e = self._todo[0]
            blocks = [e[0]] + ([e[3]] if len(e) == 6 else [])
            if sealed_only and not all(b.sealed for b in blocks):
                break
            if through is not None and not any(through is b for t in self._todo for b in ([t[0]] + ([t[3]] if len(t) == 6 else []))):
                break
            # what can fail without anything having been counted (the upload of a block) comes
            # first: the stretch stays owed; once the pass itself runs it is off the list
            views = [e[0].view(e[1], e[2])] + ([e[3].view(e[4], e[5])] if len(e) == 6 else [])
            self._todo.pop(0)
            staging_stats["runs"] += 1
            for b in blocks:
                self._before_run(b)
            if len(e) == 3:
                self._run(views[0])
            else:
                self._run_pair(views[0], views[1])

    def _before_run(self, blk) -> None:
        """QCMetrics objects that write accumulated_error_rate into the block's metas go first
        (NanoStats reads it, _qcmodule.c:5314, as the reference's driver orders its calls)"""
        for w in list(blk.writers):
            q = w()
            if q is not None and q is not self:
                q._drain(through=blk)


class _DeferringArrays(_Deferring):
    """... of a module that takes single record arrays"""

    def add_record_array(self, record_array: "FastqRecordArrayView") -> None:
        """add_record_array of every module (QCMetrics_add_record_array :2183 and its siblings)"""
        if type(record_array) is _FedArray:     # the driver loop's case, without a call: the next array of the parser's open block
            t = self._todo
            if t:
                last = t[-1]
                blk, slot = record_array._staged
                if last[0] is blk and last[2] == slot and len(last) == 3:
                    last[2] = slot + 1
                    return
        arr = _require_array(record_array)
        if not self._enqueue(arr):
            self._run(arr)


class _HostBuffer:
    """page-locked host memory (sq_host_alloc)"""

    def __init__(self, nbytes: int):
        pinned = C.c_int(0)
        self.address = lib().sq_host_alloc(nbytes, C.byref(pinned))
        if not self.address:
            raise MemoryError("out of host memory")
        self.nbytes, self.pinned = nbytes, pinned.value

    def __del__(self):
        try:
            if self.address:
                lib().sq_host_free(self.address, self.pinned)
        except Exception:
            pass
        self.address = None


def _ahead_drop() -> None:
    """what sq_batch_from_fastq_ahead sent ahead is void: the buffer it named will not be asked for"""
    if _lib._ctx is not None:
        lib().sq_ahead_drop(_lib._ctx)


class PinnedReader:
    """A binary file object over FASTQ text held in page-locked host memory (the text is copied
    there once, when the object is made).  Any parser can read() / readinto() it;
    FastqParser(..., split_on_device=True) takes the pages as they are: the upload is the only
    time the bytes move."""

    def __init__(self, data):
        view = memoryview(data).cast("B")
        self._hold = _HostBuffer(len(view))
        self._address, self._size, self._pos = self._hold.address, len(view), 0
        C.memmove(self._address, _addr(np.frombuffer(view, dtype=np.uint8)) if len(view) else 0, len(view))

    def __del__(self):
        try:
            _ahead_drop()      # before the pages go back: a block sent ahead from them must not be matched by address later
        except Exception:
            pass

    def _skip(self, n: int) -> int:
        n = min(n, self._size - self._pos)
        self._pos += n
        return n

    def readinto(self, b) -> int:
        out = memoryview(b).cast("B")
        at = self._pos
        n = self._skip(len(out))
        out[:n] = (C.c_char * n).from_address(self._address + at).raw if n else b""
        return n

    def read(self, n: int = -1) -> bytes:
        at = self._pos
        n = self._skip(self._size - self._pos if n is None or n < 0 else n)
        return C.string_at(self._address + at, n)

    def seek(self, offset: int, whence: int = 0) -> int:
        base = {0: 0, 1: self._pos, 2: self._size}[whence]
        self._pos = min(max(base + offset, 0), self._size)
        _ahead_drop()          # the next buffer is no longer the one a parser named
        return self._pos

    def tell(self) -> int:
        return self._pos

    def readable(self) -> bool:
        return True

    def seekable(self) -> bool:
        return True


class _DeviceSplitArray(FastqRecordArrayView):
    """an array whose records were split on the device.  `obj` (the buffer the reference's array
    would hold) is copied at once when it is small and fetched back from HBM, where all of it went,
    when somebody asks for it and it is big: the parser's host buffer is reused by the next array."""

    def __init__(self, batch: _DeviceBatch, address: int, nbytes: int):
        self._metas = None
        self._batch = batch
        self._writeback = None
        self._obj = C.string_at(address, nbytes) if nbytes <= (1 << 20) else None

    @property
    def obj(self):
        if self._obj is None:
            self._obj = self._batch.download()[0].tobytes()
        return self._obj

    @obj.setter
    def obj(self, v):
        self._obj = v


class FastqParser:
    """FastqParser, _qcmodule.c:889-1244: iterates record arrays over a binary
    file object, ``initial_buffersize`` bytes at a time (memchr record split on
    the host, sq_fastq_split)."""

    def __init__(self, fileobj, initial_buffersize: int = 128 * 1024, split_on_device: bool = False):
        if initial_buffersize < 1:
            raise ValueError(f"initial_buffersize must be at least 1, got {initial_buffersize}")
        self._file = fileobj
        self._read_in_size = int(initial_buffersize)
        self._leftover = b""
        self._token = _Source()      # arrays of one parser are staged together
        self._dev_left, self._dev_hold = 0, None   # split_on_device: leftover bytes, page-locked buffer
        self._feeder: Optional[_Feeder] = None
        self._source = None
        self._fast_next = None     # (sq_feeder_next, handle, byref(info), info) once the feeder reads the file itself
        self._blk: Optional[_FeedBlock] = None
        # extension: iterate with the record split done on the GPU (sq_batch_from_fastq);
        # the text is uploaded once and the metas never exist on the host unless a
        # record is indexed.  read(n) keeps the host splitter.
        self._split_on_device = bool(split_on_device)

    def __iter__(self) -> "FastqParser":
        return self

    def __next__(self) -> FastqRecordArrayView:
        fast = self._fast_next
        if fast is not None:        # the driver loop's case: the feeder reads the file itself, one foreign call per array
            rc = fast[0](fast[1], fast[4], fast[5], fast[2])
            if rc == 0:
                info = fast[3]
                n = info.n_records
                blk = self._blk
                if n and blk is not None and blk.block_id == info.block_id:
                    starts = blk.starts
                    starts.append(starts[-1] + n)
                    return _FedArray(blk, len(starts) - 2, info.byte_start, info.byte_len, info.first_record, n)
                arr = self._fed_array(info)      # a new block, or the end of the file
            else:
                check(rc)
                arr = self._create(1, sys.maxsize)   # (SQ_FEED_MORE cannot come from a feeder with a source: kept for safety)
        else:
            arr = self._create_on_device() if self._split_on_device else self._create(1, sys.maxsize)
        if len(arr) == 0:
            raise StopIteration
        return arr

    def read(self, number_of_records: int) -> FastqRecordArrayView:
        if number_of_records < 1:
            raise ValueError(f"number_of_records should be greater than 1, got {number_of_records}")
        return self._create(number_of_records, number_of_records)

    def _create(self, min_records: int, max_records: int) -> FastqRecordArrayView:
        if _USE_FEEDER and _STAGE_LIMIT > 0:
            return self._create_fed(min_records, max_records)
        return self._create_py(min_records, max_records)

    def _create_fed(self, min_records: int, max_records: int) -> FastqRecordArrayView:
        """FastqParser_create_record_array (_qcmodule.c:964-1184) by sq_feeder_next: the file's bytes
        go straight into a pinned staging block, the array is a window of it (csrc/sq_feed.hip)"""
        f = self._feeder
        if f is None:
            f = self._feeder = _Feeder(self._read_in_size)
            # the feeder may read the file by itself (worker threads).  What they read from belongs to the FEEDER object: it must
            # outlive sq_feeder_free, which waits for them (a parser dropped in the middle of a file let go of the BytesIO's bytes
            # first, now and then, and a worker copied from freed memory)
            self._source = f.source = _feeder_source(f, self._file)
            if self._source is not None:
                info = _FeedArrayInfo()
                self._fast_next = (lib().sq_feeder_next, f.h, C.byref(info), info, C.c_size_t(1), C.c_size_t(_MAX_RECORDS))
        info = _FeedArrayInfo()
        room = C.c_size_t(0)
        while True:
            rc = lib().sq_feeder_next(f.h, min_records, min(max_records, (1 << 63) - 1), C.byref(info))
            if rc != 1:
                break
            p = lib().sq_feeder_fill(f.h, C.byref(room))
            if not p:
                raise MemoryError(_lib.last_error())
            got = self._file.readinto((C.c_char * room.value).from_address(p)) or 0
            check(lib().sq_feeder_filled(f.h, got))
        check(rc)
        return self._fed_array(info)

    def _fed_array(self, info) -> FastqRecordArrayView:
        f = self._feeder
        if info.n_records == 0:
            return FastqRecordArrayView._from_buffer(b"", np.zeros(0, dtype=META_DTYPE))
        blk = self._blk
        if blk is None or blk.block_id != info.block_id:
            if blk is not None and not blk.closed:     # the feeder went on to a new block by itself: the old one is complete
                blk.closed = True
                f.retire(blk)
            blk = self._blk = _FeedBlock(self._token, f, info.block_id)
        slot = blk.add(info.n_records)
        return _FedArray(blk, slot, info.byte_start, info.byte_len, info.first_record, info.n_records)

    def _create_on_device(self) -> FastqRecordArrayView:
        try:
            arr = self._create_on_device_inner()
        except BaseException:
            _ahead_drop()      # the parse is over: whatever was sent ahead has no taker
            raise
        if len(arr) == 0:
            _ahead_drop()
        return arr

    def _create_on_device_inner(self) -> FastqRecordArrayView:
        """The loop of FastqParser_create_record_array (_qcmodule.c:964-1184) with the
        ASCII check and the record split done by sq_batch_from_fastq.  The buffer is page-locked
        (the upload runs at the bus rate); a PinnedReader hands its own pages over, no copy at all."""
        src = self._file if isinstance(self._file, PinnedReader) else None
        left = self._dev_left            # bytes of the previous buffer no complete record covered
        first, eof = True, False
        have = left
        while True:
            want = max(self._read_in_size - have, 0) if first else self._read_in_size
            first = False
            if want > 0:
                if src is not None:
                    got = src._skip(want)
                else:
                    buf = self._dev_buffer(have + want)
                    got = self._file.readinto((C.c_char * want).from_address(buf + have)) or 0
                if got:
                    have += got
                else:
                    eof = True
            if have == 0:
                self._dev_left = 0
                return FastqRecordArrayView._from_buffer(b"", np.zeros(0, dtype=META_DTYPE))
            addr = src._address + src._pos - have if src is not None else self._dev_buffer(have)
            if eof and C.string_at(addr, have).count(b"\n") < 4:  # :1073-1081
                raise EOFError("Incomplete record at the end of file " + C.string_at(addr, have).decode("latin-1"))
            consumed = C.c_size_t(0)
            if src is not None:    # the pages of the next buffer are known already: their upload starts now
                # (with the end of this buffer: the leftover of this call will be in front of the next one)
                back = min(src._pos, have, 64 << 10)
                nxt = min(self._read_in_size, src._size - src._pos) if os.environ.get("SQ_AHEAD", "1") != "0" else 0
                h = lib().sq_batch_from_fastq_ahead(context(), addr, have, C.byref(consumed),
                                                    src._address + src._pos - back if nxt > 0 else None, nxt + back if nxt > 0 else 0)
            else:
                h = lib().sq_batch_from_fastq(context(), addr, have, C.byref(consumed))
            if not h:
                raise ValueError(_lib.last_error())
            batch = _DeviceBatch(h)
            if batch.number_of_records >= 1:
                break
            if eof:
                raise EOFError("Incomplete record at the end of file " + C.string_at(addr, have).decode("latin-1"))
        arr = _DeviceSplitArray(batch, addr, have)
        arr._source = self._token
        self._dev_left = have - consumed.value
        if src is None and self._dev_left:     # the leftover moves to the front of the buffer
            C.memmove(self._dev_buffer(self._dev_left), addr + consumed.value, self._dev_left)
        return arr

    def _dev_buffer(self, nbytes: int) -> int:
        """address of the parser's page-locked buffer, at least nbytes long (contents kept)"""
        hold = self._dev_hold
        if hold is None or hold.nbytes < nbytes:
            new = _HostBuffer(max(nbytes, 2 * (hold.nbytes if hold else 0)))
            if hold is not None:
                C.memmove(new.address, hold.address, hold.nbytes)
            hold = self._dev_hold = new
        return hold.address

    def _create_py(self, min_records: int, max_records: int) -> FastqRecordArrayView:
        """FastqParser_create_record_array, _qcmodule.c:964-1184: a new buffer of
        ``initial_buffersize`` bytes seeded with the leftover of the previous call,
        enlarged by the same amount until ``min_records`` records fit."""
        buf = bytearray(self._leftover)
        first, eof = True, False
        metas = np.zeros(0, dtype=META_DTYPE)
        consumed = 0
        while True:
            want = max(self._read_in_size - len(buf), 0) if first else self._read_in_size
            first = False
            if want > 0:
                chunk = bytearray(want)
                got = self._file.readinto(chunk) or 0
                if got:
                    new = bytes(chunk[:got])
                    bad = lib().sq_first_non_ascii(_addr(new), got)
                    if bad >= 0:  # :1055-1067
                        raise ValueError(f"Found non-ASCII character in file: {chr(new[bad])}")
                    buf += new
                else:
                    eof = True
            if len(buf) == 0:
                break  # :1069 entire file is read
            if eof and buf.count(b"\n") < 4:  # :1073-1081
                raise EOFError("Incomplete record at the end of file " + bytes(buf).decode("latin-1"))
            cap = len(buf) // 64 + 16
            while True:
                tmp = np.zeros(cap, dtype=META_DTYPE)
                c = C.c_size_t(0)
                view = np.frombuffer(buf, dtype=np.uint8)
                n = lib().sq_fastq_split(view.ctypes.data, len(buf), tmp.ctypes.data,
                                         min(cap, max_records), C.byref(c))
                del view
                check(n)
                if n == cap and cap < max_records:
                    cap *= 4
                    continue
                break
            metas, consumed = tmp[:n].copy(), c.value
            if n >= min_records:
                break
            if eof:
                if n == 0:
                    raise EOFError("Incomplete record at the end of file " + bytes(buf).decode("latin-1"))
                break
        obj = bytes(buf)
        self._leftover = obj[consumed:]
        arr = FastqRecordArrayView._from_buffer(obj, metas)
        arr._source = self._token   # arrays of one parser are staged together
        return arr


class BamParser:
    """BamParser, _qcmodule.c:1362-1722: iterates record arrays over an *uncompressed* BAM
    stream (the caller removes BGZF, as the reference's xopen does).  The record walk runs
    on the host (sq_bam_scan), the decode on the GPU (sq_batch_from_bam); the
    arrays live in HBM and come to the host only when a record is looked at.  Secondary and
    supplementary alignments are left out (:1262)."""

    def __init__(self, fileobj, initial_buffersize: int = 48 * 1024):
        if initial_buffersize < 4:
            raise ValueError(f"initial_buffersize must be at least 4, got {initial_buffersize}")
        magic = fileobj.read(8)
        if type(magic) is not bytes:
            raise TypeError(f"file_obj {fileobj!r} is not a binary IO type, got {type(fileobj)!r}")
        if len(magic) < 8:
            raise EOFError("Truncated BAM file")
        if magic[:4] != b"BAM\x01":
            raise ValueError(f"fileobj: {fileobj!r}, is not a BAM file. No BAM magic, "
                             f"instead found: {magic!r}")
        l_text = int.from_bytes(magic[4:8], "little")
        header = fileobj.read(l_text)
        if len(header) != l_text:
            raise EOFError("Truncated BAM file")
        n_ref = fileobj.read(4)
        if len(n_ref) != 4:
            raise EOFError("Truncated BAM file")
        for _ in range(int.from_bytes(n_ref, "little")):
            l_name = fileobj.read(4)
            if len(l_name) != 4:
                raise EOFError("Truncated BAM file")
            want = int.from_bytes(l_name, "little") + 4  # name and l_ref
            if len(fileobj.read(want)) != want:
                raise EOFError("Truncated BAM file")
        self.header = header
        self._file = fileobj
        self._read_in_size = int(initial_buffersize)
        self._leftover = b""
        self._token = _Source()      # arrays of one parser are staged together
        self._dev_left, self._dev_hold = 0, None   # split_on_device: leftover bytes, page-locked buffer
        self._feeder: Optional[_Feeder] = None
        self._blk: Optional[_FeedBlock] = None

    def __iter__(self) -> "BamParser":
        return self

    def __next__(self) -> FastqRecordArrayView:
        """BamParser__next__ :1506-1703: the buffer grows until it holds one complete record"""
        buf = bytearray(self._leftover)
        while True:
            if len(buf) >= 4:  # :1527-1531 enough for the record in front
                want = max(int.from_bytes(buf[:4], "little"), self._read_in_size)
            else:
                want = self._read_in_size - len(buf)
            chunk = bytearray(want)
            got = self._file.readinto(chunk) or 0
            if len(buf) + got == 0:
                raise StopIteration  # :1564
            if got == 0:  # :1569-1577
                raise EOFError(f"Incomplete record at the end of file {bytes(buf)!r}")
            buf += chunk[:got]
            view = np.frombuffer(buf, dtype=np.uint8)
            consumed, skipped = C.c_size_t(0), C.c_uint64(0)
            n = check(lib().sq_bam_scan(view.ctypes.data, len(buf), None, 0, C.byref(consumed), C.byref(skipped)))
            if n + skipped.value == 0:
                del view
                continue
            offsets = np.zeros(max(n, 1), dtype=np.uint64)
            check(lib().sq_bam_scan(view.ctypes.data, len(buf), offsets.ctypes.data, n, C.byref(consumed),
                                    C.byref(skipped)))
            h = lib().sq_batch_from_bam(context(), view.ctypes.data, consumed.value, offsets.ctypes.data, n)
            del view
            if not h:
                raise MemoryError(_lib.last_error())
            self._leftover = bytes(buf[consumed.value:])
            return FastqRecordArrayView._from_device(_DeviceBatch(h))


# ---------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------
def _u64_array(fn, handle, n_hint: Optional[int] = None) -> array.array:
    n = check(fn(handle, None, 0))
    out = np.zeros(n, dtype=np.uint64)
    if n:
        check(fn(handle, out.ctypes.data, n))
    a = array.array("Q")
    a.frombytes(out.tobytes())
    return a


class QCMetrics(_DeferringArrays):
    """_qcmodule.c:1786-2385"""

    def __init__(self, end_anchor_length: int = DEFAULT_END_ANCHOR_LENGTH):
        if end_anchor_length < 0 or end_anchor_length > 0xFFFFFFFF:
            raise ValueError(f"end_anchor_length must be between 0 and {0xFFFFFFFF}, "
                             f"got {end_anchor_length}")
        h = lib().sq_qcmetrics_new(context(), end_anchor_length)
        if not h:
            raise MemoryError(_lib.last_error())
        self._init_defer(h)
        self._pending: List[FastqRecordArrayView] = []
        self._polled: Optional[int] = None     # arrays of _pending in front of the armed poll (_settle)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_qcmetrics_free(self._handle)
        except Exception:
            pass

    def _staged_in(self, blk, arr) -> None:
        arr._writeback = weakref.ref(self)
        if not any(w() is self for w in blk.writers):
            blk.writers.append(weakref.ref(self))

    def _run(self, arr: FastqRecordArrayView) -> None:
        check(lib().sq_qcmetrics_add_batch(self._handle, arr._device().handle))
        self._track(arr)

    def _track(self, arr: FastqRecordArrayView) -> None:
        arr._writeback = weakref.ref(self)
        self._pending.append(arr)
        if len(self._pending) > 64:
            self.flush()
        elif len(self._pending) > 1:
            self._settle()

    def _settle(self) -> None:
        """lets go of the arrays whose passes have ended without an invalid phred character -- found out without waiting
        (sq_qcmetrics_poll).  A pending array holds its staging block's copy in HBM; until round 6 every block of a file
        stayed there until the 65th came or somebody flushed, and each one was a fresh hipMalloc (exp_e2e_pool.txt)"""
        r = lib().sq_qcmetrics_poll(self._handle)
        if self._polled is None:
            if r == 0:
                self._polled = len(self._pending)    # armed behind the passes of that many arrays
            return
        if r == 0:
            return
        n, self._polled = self._polled, None
        if r == 1:
            done, self._pending = self._pending[:n], self._pending[n:]
            self._written_back(done)
            if self._pending and lib().sq_qcmetrics_poll(self._handle) == 0:
                self._polled = len(self._pending)

    def add_read(self, read: FastqRecordView) -> None:
        view = _require_view(read)
        arr = FastqRecordArrayView([view])
        self._add_now(arr)
        self.flush()
        view._meta["accumulated_error_rate"] = arr._metas["accumulated_error_rate"]

    def flush(self) -> None:
        """Waits for the enqueued passes, writes accumulated_error_rate back into the
        arrays that went through (:2126) and raises a deferred ValueError.

        An invalid phred character (:2073-2075, 2102-2105): the reference raises inside the call
        that brought the array, with the reads in front of the offending one counted, the
        offender's bases and the phred counts in front of the character too, and nothing else.
        The passes here run whole batches and flag the read; the flag is found at this point.
        Every array since the last flush is then asked for its first offending record and has
        its tail taken back (sq_qcmetrics_uncount_tail), so the tables are what the reference's
        would be had each of those calls raised; the error of the first one is raised, and the
        object stays usable."""
        self._drain()
        pending, self._pending = self._pending, []
        rc = lib().sq_qcmetrics_flush(self._handle)
        if self._polled is not None:     # an armed poll ends here (the passes have ended: this call disarms it)
            lib().sq_qcmetrics_poll(self._handle)
            self._polled = None
        error = None
        if rc < 0:
            msg = _lib.last_error()
            for arr in pending:
                if arr._batch is None or not len(arr):
                    continue
                # one stretch per call that brought records of this batch (a staging block holds many)
                calls = [(r0, r1) for _, r0, r1 in getattr(arr, "_children", ())] or [(0, len(arr))]
                keep = None
                for r0, r1 in calls:
                    idx = lib().sq_batch_first_invalid_phred(arr._batch.handle, r0, r1)
                    if idx < 0:
                        continue
                    if keep is None:
                        keep = np.ones(len(arr), dtype=bool)
                    keep[idx + 1:r1] = False
                    lengths = arr._host_metas()["sequence_length"]
                    kept_max = int(lengths[keep].max()) if keep.any() else 0
                    check(lib().sq_qcmetrics_uncount_tail(self._handle, arr._batch.handle, idx, r1, kept_max))
                    if error is None:
                        view = arr[int(idx)]
                        q = view._slice("qualities_offset", int(view._meta[0]["sequence_length"]))
                        bad = next((c for c in q if not 33 <= c <= 33 + PHRED_MAX), ord("?"))
                        error = ValueError(f"Not a valid phred character: {chr(bad)}")
            if error is None:
                error = ValueError(msg)
            check(lib().sq_qcmetrics_flush(self._handle))    # the log of the handled batches ends here
        self._written_back(pending)
        if error is not None:
            raise error

    @staticmethod
    def _written_back(arrays) -> None:
        """accumulated_error_rate of arrays whose passes have ended (:2126)"""
        for arr in arrays:
            blk = getattr(arr, "_blk", None)
            blk = blk() if blk is not None else None
            if blk is not None:
                blk.rates_version += 1      # its arrays fetch the rates from HBM when somebody asks for them
            if arr._metas is not None and arr._batch is not None and len(arr._metas):
                rates = arr._batch.error_rates()
                arr._metas["accumulated_error_rate"] = rates
                for ref, r0, r1 in getattr(arr, "_children", ()):   # the arrays a staging block was made of
                    child = ref() if ref is not None else None
                    if child is not None:
                        child._metas["accumulated_error_rate"] = rates[r0:r1]
                        child._writeback = None
            arr._writeback = None

    @property
    def number_of_reads(self) -> int:
        return lib().sq_qcmetrics_number_of_reads(self._h)

    @property
    def max_length(self) -> int:
        return lib().sq_qcmetrics_max_length(self._h)

    @property
    def end_anchor_length(self) -> int:
        return lib().sq_qcmetrics_end_anchor_length(self._h)

    def _table(self, fn) -> array.array:
        self.flush()
        return _u64_array(fn, self._h)

    def base_count_table(self) -> array.array:
        return self._table(lib().sq_qcmetrics_base_count_table)

    def phred_count_table(self) -> array.array:
        return self._table(lib().sq_qcmetrics_phred_count_table)

    def end_anchored_base_count_table(self) -> array.array:
        return self._table(lib().sq_qcmetrics_end_anchored_base_count_table)

    def end_anchored_phred_count_table(self) -> array.array:
        return self._table(lib().sq_qcmetrics_end_anchored_phred_count_table)

    def gc_content(self) -> array.array:
        return self._table(lib().sq_qcmetrics_gc_content)

    def phred_scores(self) -> array.array:
        return self._table(lib().sq_qcmetrics_phred_scores)


class AdapterCounter(_DeferringArrays):
    """_qcmodule.c:2391-2969"""

    def __init__(self, adapters: Iterable[str]):
        try:
            adapters = tuple(adapters)
        except TypeError:
            raise TypeError(f"{type(adapters).__name__!r} object is not iterable")
        if len(adapters) < 1:
            raise ValueError("At least one adapter is expected")
        for a in adapters:
            if type(a) is not str:
                raise TypeError("All adapter sequences must be of type str, "
                                f"got {type(a)!r}, for {a!r}")
            if not a.isascii():
                raise ValueError(f"Adapter must contain only ASCII characters: {a!r}")
            if len(a) > MAX_SEQUENCE_SIZE:
                raise ValueError(f"Maximum adapter size is {MAX_SEQUENCE_SIZE}, "
                                 f"got {len(a)} for {a!r}")
        self.adapters = adapters
        enc = [a.encode("ascii") for a in adapters]
        ptrs = (C.c_char_p * len(enc))(*enc)
        lens = (C.c_size_t * len(enc))(*[len(e) for e in enc])
        h = lib().sq_adaptercounter_new(context(), C.cast(ptrs, C.c_void_p),
                                        C.cast(lens, C.c_void_p), len(enc))
        if not h:
            raise ValueError(_lib.last_error())
        self._init_defer(h)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_adaptercounter_free(self._handle)
        except Exception:
            pass

    def _run(self, arr: FastqRecordArrayView) -> None:
        check(lib().sq_adaptercounter_add_batch(self._handle, arr._device().handle))

    def add_read(self, read: FastqRecordView) -> None:
        self._add_now(FastqRecordArrayView([_require_view(read)]))

    def flush(self) -> None:
        check(lib().sq_adaptercounter_flush(self._h))

    @property
    def number_of_sequences(self) -> int:
        return lib().sq_adaptercounter_number_of_sequences(self._h)

    @property
    def max_length(self) -> int:
        return lib().sq_adaptercounter_max_length(self._h)

    def get_counts(self) -> List[Tuple[str, array.array, array.array]]:
        ml = self.max_length
        out = []
        for i, a in enumerate(self.adapters):
            f = np.zeros(ml, dtype=np.uint64)
            r = np.zeros(ml, dtype=np.uint64)
            check(lib().sq_adaptercounter_get_counts(self._h, i, f.ctypes.data, r.ctypes.data, ml))
            fa, ra = array.array("Q"), array.array("Q")
            fa.frombytes(f.tobytes())
            ra.frombytes(r.tobytes())
            out.append((a, fa, ra))
        return out


class PerTileQuality(_DeferringArrays):
    """_qcmodule.c:2975-3397"""

    def __init__(self):
        h = lib().sq_pertile_new(context())
        if not h:
            raise MemoryError(_lib.last_error())
        self._init_defer(h)
        self._off = False   # skipped for good (:3137-3148), as of the work that has run

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_pertile_free(self._handle)
        except Exception:
            pass

    def add_record_array(self, record_array: FastqRecordArrayView) -> None:
        arr = _require_array(record_array)
        if self._off:      # :3126
            return
        if not self._enqueue(arr):
            self._run(arr)

    def _run(self, arr: FastqRecordArrayView) -> None:
        if self._off:
            return
        check(lib().sq_pertile_add_batch(self._handle, arr._device().handle))
        self._off = lib().sq_pertile_skipped_reason(self._handle) is not None

    def add_read(self, read: FastqRecordView) -> None:
        self._add_now(FastqRecordArrayView([_require_view(read)]))

    def flush(self) -> None:
        check(lib().sq_pertile_flush(self._h))

    @property
    def number_of_reads(self) -> int:
        return lib().sq_pertile_number_of_reads(self._h)

    @property
    def max_length(self) -> int:
        return lib().sq_pertile_max_length(self._h)

    @property
    def skipped_reason(self) -> Optional[str]:
        r = lib().sq_pertile_skipped_reason(self._h)
        return None if r is None else r.decode("ascii", "replace")

    def get_tile_counts(self) -> List[Tuple[int, List[float], List[int]]]:
        nt = lib().sq_pertile_number_of_tiles(self._h)
        ml = self.max_length
        ids = np.zeros(nt, dtype=np.int64)
        err = np.zeros((nt, ml), dtype=np.float64)
        cnt = np.zeros((nt, ml), dtype=np.uint64)
        if nt:
            check(lib().sq_pertile_get_tile_counts(self._h, ids.ctypes.data, err.ctypes.data,
                                                   cnt.ctypes.data, nt, ml))
        return [(int(ids[i]), err[i].tolist(), [int(x) for x in cnt[i]]) for i in range(nt)]


class FusedPass(_DeferringArrays):
    """One pass over each record array for any of QCMetrics / AdapterCounter /
    PerTileQuality (sq_fused_add_batch): the same results as calling the three
    add_record_array methods in turn, with the records read from HBM once."""

    def __init__(self, qc_metrics: Optional[QCMetrics] = None,
                 adapter_counter: Optional[AdapterCounter] = None,
                 per_tile_quality: Optional[PerTileQuality] = None):
        self.qc_metrics = qc_metrics
        self.adapter_counter = adapter_counter
        self.per_tile_quality = per_tile_quality
        self._init_defer(None)
        for mod in (qc_metrics, adapter_counter, per_tile_quality):
            if mod is not None:
                mod._upstream.append(self)   # the module's getters run this pass first

    def _staged_in(self, blk, arr) -> None:
        if self.qc_metrics is not None:
            self.qc_metrics._staged_in(blk, arr)

    def _drain_own(self, sealed_only: bool = False, through=None) -> None:
        # what the modules owe from calls of their own goes first
        for mod in (self.qc_metrics, self.adapter_counter, self.per_tile_quality):
            if mod is not None:
                mod._drain_own(sealed_only, through)
        _Deferring._drain_own(self, sealed_only, through)

    def _run(self, arr: FastqRecordArrayView) -> None:
        m, a, p = self.qc_metrics, self.adapter_counter, self.per_tile_quality
        check(lib().sq_fused_add_batch(arr._device().handle, m._handle if m else None,
                                       a._handle if a else None, p._handle if p else None))
        if p is not None:
            p._off = lib().sq_pertile_skipped_reason(p._handle) is not None
        if m is not None:
            m._track(arr)


class PairedPass(_Deferring):
    """One call per pair of record arrays for what the reference's driver does with it (__main__.py:279-306):
    QCMetrics + PerTileQuality on read 1, the same on read 2, InsertSizeMetrics on the pair
    (sq_paired_add_batches) -- the results of the five add_record_array[_pair] calls in that order.  Arrays of
    one read length each take two passes over the records instead of seven (csrc/sq_pair.hip)."""

    def __init__(self, qc_metrics1: Optional[QCMetrics] = None, per_tile_quality1: Optional[PerTileQuality] = None,
                 qc_metrics2: Optional[QCMetrics] = None, per_tile_quality2: Optional[PerTileQuality] = None,
                 insert_size_metrics: Optional["InsertSizeMetrics"] = None):
        self.qc_metrics1, self.per_tile_quality1 = qc_metrics1, per_tile_quality1
        self.qc_metrics2, self.per_tile_quality2 = qc_metrics2, per_tile_quality2
        self.insert_size_metrics = insert_size_metrics
        self._init_defer(None)
        for mod in self._modules():
            mod._upstream.append(self)   # the module's getters run this pass first

    def _modules(self):
        return [m for m in (self.qc_metrics1, self.per_tile_quality1, self.qc_metrics2, self.per_tile_quality2,
                            self.insert_size_metrics) if m is not None]

    def add_record_array_pair(self, record_array1, record_array2) -> None:
        a1 = _require_array(record_array1, "record_array1")
        a2 = _require_array(record_array2, "record_array2")
        if len(a1) != len(a2):   # InsertSizeMetrics_add_record_array_pair :5842-5848
            raise ValueError("record_array1 and record_array2 must be of the same size. "
                             f"Got {len(a1)} and {len(a2)} respectively.")
        if self._enqueue_pair(a1, a2):
            for m, a in ((self.qc_metrics1, a1), (self.qc_metrics2, a2)):
                if m is not None:
                    m._staged_in(a._staged[0], a)
        else:
            self._run_pair(a1, a2)

    def _drain_own(self, sealed_only: bool = False, through=None) -> None:
        for mod in self._modules():      # what the modules owe from calls of their own goes first
            mod._drain_own(sealed_only, through)
        _Deferring._drain_own(self, sealed_only, through)

    def _run_pair(self, a1: FastqRecordArrayView, a2: FastqRecordArrayView) -> None:
        def h(mod):
            return mod._handle if mod is not None else None
        p1, p2 = self.per_tile_quality1, self.per_tile_quality2
        if p1 is not None and p1._off:
            p1 = None            # :3126: the module has stopped
        if p2 is not None and p2._off:
            p2 = None
        check(lib().sq_paired_add_batches(a1._device().handle, a2._device().handle, h(self.qc_metrics1), h(p1),
                                          h(self.qc_metrics2), h(p2), h(self.insert_size_metrics)))
        for p in (p1, p2):
            if p is not None:
                p._off = lib().sq_pertile_skipped_reason(p._handle) is not None
        for m, a in ((self.qc_metrics1, a1), (self.qc_metrics2, a2)):
            if m is not None:
                m._track(a)


def _kmer_to_sequence(kmer: int, k: int) -> str:
    """kmer_to_sequence, _qcmodule.c:3405-3414"""
    return "".join("ACGT"[(kmer >> (2 * (k - 1 - i))) & 3] for i in range(k))


class OverrepresentedSequences(_DeferringArrays):
    """_qcmodule.c:3435-4236"""

    def __init__(self, max_unique_fragments: int = DEFAULT_MAX_UNIQUE_FRAGMENTS,
                 fragment_length: int = DEFAULT_FRAGMENT_LENGTH,
                 sample_every: int = DEFAULT_UNIQUE_SAMPLE_EVERY,
                 bases_from_start: int = DEFAULT_BASES_FROM_START,
                 bases_from_end: int = DEFAULT_BASES_FROM_END):
        h = lib().sq_overrep_new(context(), max_unique_fragments, fragment_length,
                                 sample_every, bases_from_start, bases_from_end)
        if not h:
            raise ValueError(_lib.last_error())
        self._init_defer(h)
        self.max_unique_fragments = max_unique_fragments
        self.fragment_length = fragment_length
        self.sample_every = sample_every
        self._warned = 0
        self._first_record = 0

    def set_shard(self, first_record_index: int) -> None:
        """This object sees records [first_record_index, ...) of a job that other ranks
        share (sequali_amd.dist.merge_overrepresented joins them)."""
        check(lib().sq_overrep_set_shard(self._h, first_record_index))
        self._first_record = first_record_index

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_overrep_free(self._handle)
        except Exception:
            pass

    def _run(self, arr: FastqRecordArrayView) -> None:
        before = self._first_record + lib().sq_overrep_number_of_sequences(self._handle)
        check(lib().sq_overrep_add_batch(self._handle, arr._device().handle))
        count = lib().sq_overrep_warning_count(self._handle)
        if count != self._warned:  # :3931-3938, once per array here
            self._warned = count
            idx = lib().sq_overrep_last_warning_record(self._handle) - before
            culprit = arr[idx].sequence() if arr._metas is not None and 0 <= idx < len(arr) else "?"
            warnings.warn("Sequence contains a chacter that is not A, C, G, T or N: "
                          f"{culprit!r}", UserWarning, stacklevel=2)

    def add_read(self, read: FastqRecordView) -> None:
        self._add_now(FastqRecordArrayView([_require_view(read)]))

    def flush(self) -> None:
        check(lib().sq_overrep_flush(self._h))

    @property
    def number_of_sequences(self) -> int:
        return lib().sq_overrep_number_of_sequences(self._h)

    @property
    def sampled_sequences(self) -> int:
        return lib().sq_overrep_sampled_sequences(self._h)

    @property
    def collected_unique_fragments(self) -> int:
        return lib().sq_overrep_collected_unique_fragments(self._h)

    @property
    def total_fragments(self) -> int:
        return lib().sq_overrep_total_fragments(self._h)

    def _counts(self) -> Tuple[np.ndarray, np.ndarray]:
        n = check(lib().sq_overrep_get_counts(self._h, None, None, 0))
        km = np.zeros(n, dtype=np.uint64)
        ct = np.zeros(n, dtype=np.uint64)
        if n:
            check(lib().sq_overrep_get_counts(self._h, km.ctypes.data, ct.ctypes.data, n))
        return km, ct

    def sequence_counts(self) -> Dict[str, int]:
        km, ct = self._counts()
        k = self.fragment_length
        return {_kmer_to_sequence(int(a), k): int(b) for a, b in zip(km, ct)}

    def overrepresented_sequences(self, threshold_fraction: float = 0.0001,
                                  min_threshold: int = 1,
                                  max_threshold: int = sys.maxsize
                                  ) -> List[Tuple[int, float, str]]:
        """:4091-4180"""
        import math
        if threshold_fraction < 0.0 or threshold_fraction > 1.0:
            raise ValueError("threshold_fraction must be between 0.0 and 1.0 got, "
                             f"{threshold_fraction!r}")
        if min_threshold < 1:
            raise ValueError(f"min_threshold must be at least 1, got {min_threshold}")
        if max_threshold < 1:
            raise ValueError(f"max_threshold must be at least 1, got {max_threshold}")
        if max_threshold < min_threshold:
            raise ValueError(f"max_threshold ({max_threshold}) must be greater than "
                             f"min_threshold ({min_threshold})")
        sampled = self.sampled_sequences
        hits = min(max_threshold, max(min_threshold, math.ceil(threshold_fraction * sampled)))
        km, ct = self._counts()
        k = self.fragment_length
        res = [(int(c), int(c) / sampled, _kmer_to_sequence(int(a), k))
               for a, c in zip(km, ct) if c >= hits]
        res.sort(reverse=True)
        return res


class DedupEstimator(_DeferringArrays):
    """_qcmodule.c:4270-4802"""

    def __init__(self, max_stored_fingerprints: int = DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS, *,
                 front_sequence_length: int = DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH,
                 back_sequence_length: int = DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH,
                 front_sequence_offset: int = DEFAULT_FINGERPRINT_FRONT_SEQUENCE_OFFSET,
                 back_sequence_offset: int = DEFAULT_FINGERPRINT_BACK_SEQUENCE_OFFSET):
        h = lib().sq_dedup_new(context(), max_stored_fingerprints, front_sequence_length,
                               back_sequence_length, front_sequence_offset,
                               back_sequence_offset)
        if not h:
            raise ValueError(_lib.last_error())
        self._init_defer(h)
        self.front_sequence_length = front_sequence_length
        self.back_sequence_length = back_sequence_length
        self.front_sequence_offset = front_sequence_offset
        self.back_sequence_offset = back_sequence_offset

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_dedup_free(self._handle)
        except Exception:
            pass

    def set_deferred(self, on: bool = True) -> None:
        """Shard of a multi-rank job: only hash now, insert when the shard in front is done
        (sequali_amd.dist.merge_dedup)."""
        check(lib().sq_dedup_set_deferred(self._h, 1 if on else 0))

    def _run(self, arr: FastqRecordArrayView) -> None:
        check(lib().sq_dedup_add_batch(self._handle, arr._device().handle))

    def _run_pair(self, a1: FastqRecordArrayView, a2: FastqRecordArrayView) -> None:
        check(lib().sq_dedup_add_batch_pair(self._handle, a1._device().handle, a2._device().handle))

    def add_record_array_pair(self, record_array1, record_array2) -> None:
        a1 = _require_array(record_array1, "record_array1")
        a2 = _require_array(record_array2, "record_array2")
        if len(a1) != len(a2):
            raise ValueError("record_array1 and record_array2 must be of the same size. "
                             f"Got {len(a1)} and {len(a2)} respectively.")
        if not self._enqueue_pair(a1, a2):
            self._run_pair(a1, a2)

    def add_sequence(self, sequence: str) -> None:
        if type(sequence) is not str:
            raise TypeError(f"sequence should be a str object, got {type(sequence)!r}")
        if not sequence.isascii():
            raise ValueError("sequence should consist only of ASCII characters.")
        self._add_now(_array_of_sequences([sequence]))

    def add_sequence_pair(self, sequence1: str, sequence2: str) -> None:
        for s in (sequence1, sequence2):
            if not isinstance(s, str):
                raise TypeError(f"add_sequence_pair() argument must be str, not {type(s).__name__}")
            if not s.isascii():
                raise ValueError("sequence should consist only of ASCII characters.")
        self._add_pair_now(_array_of_sequences([sequence1]), _array_of_sequences([sequence2]))

    @property
    def _modulo_bits(self) -> int:
        return lib().sq_dedup_modulo_bits(self._h)

    @property
    def _hash_table_size(self) -> int:
        return lib().sq_dedup_hash_table_size(self._h)

    @property
    def tracked_sequences(self) -> int:
        return lib().sq_dedup_tracked_sequences(self._h)

    def duplication_counts(self) -> array.array:
        return _u64_array(lib().sq_dedup_duplication_counts, self._h)


class InsertSizeMetrics(_Deferring):
    """_qcmodule.c:5456-5982"""

    def __init__(self, max_adapters: int = 10000):
        h = lib().sq_insertsize_new(context(), max_adapters)
        if not h:
            raise ValueError(_lib.last_error())
        self._init_defer(h)

    def _run_pair(self, a1: FastqRecordArrayView, a2: FastqRecordArrayView) -> None:
        check(lib().sq_insertsize_add_batch_pair(self._handle, a1._device().handle, a2._device().handle))

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_insertsize_free(self._handle)
        except Exception:
            pass

    def add_record_array_pair(self, record_array1, record_array2) -> None:
        a1 = _require_array(record_array1, "record_array1")
        a2 = _require_array(record_array2, "record_array2")
        if len(a1) != len(a2):
            raise ValueError("record_array1 and record_array2 must be of the same size. "
                             f"Got {len(a1)} and {len(a2)} respectively.")
        if not self._enqueue_pair(a1, a2):
            self._run_pair(a1, a2)

    def set_shard(self, first_pair_index: int, table_bits: int = 22) -> None:
        """This object sees pairs [first_pair_index, ...) of a job that other ranks share
        (sequali_amd.dist.merge_insertsize joins them)."""
        check(lib().sq_insertsize_set_shard(self._h, first_pair_index, table_bits))

    def add_sequence_pair(self, sequence1: str, sequence2: str) -> None:
        for label, s in (("sequence1", sequence1), ("sequence2", sequence2)):
            if not isinstance(s, str):
                raise TypeError(f"add_sequence_pair() argument must be str, not {type(s).__name__}")
            if not s.isascii():
                raise ValueError(f"{label} should consist only of ASCII characters.")
        self._add_pair_now(_array_of_sequences([sequence1]), _array_of_sequences([sequence2]))

    @property
    def total_reads(self) -> int:
        return lib().sq_insertsize_total_reads(self._h)

    @property
    def number_of_adapters_read1(self) -> int:
        return lib().sq_insertsize_number_of_adapters_read1(self._h)

    @property
    def number_of_adapters_read2(self) -> int:
        return lib().sq_insertsize_number_of_adapters_read2(self._h)

    def insert_sizes(self) -> array.array:
        return _u64_array(lib().sq_insertsize_insert_sizes, self._h)

    def _adapters(self, read2: int) -> List[Tuple[str, int]]:
        n = check(lib().sq_insertsize_adapters(self._h, read2, None, None, None, 0))
        by = np.zeros((n, INSERT_SIZE_MAX_ADAPTER_STORE_SIZE), dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint8)
        ct = np.zeros(n, dtype=np.uint64)
        if n:
            check(lib().sq_insertsize_adapters(self._h, read2, by.ctypes.data, ln.ctypes.data,
                                               ct.ctypes.data, n))
        return [(bytes(by[i, :ln[i]]).decode("ascii"), int(ct[i])) for i in range(n)]

    def adapters_read1(self) -> List[Tuple[str, int]]:
        return self._adapters(0)

    def adapters_read2(self) -> List[Tuple[str, int]]:
        return self._adapters(1)


NANOINFO_DTYPE = np.dtype([("start_time", "<i8"), ("duration", "<f4"), ("channel_id", "<i4"),
                           ("length", "<u4"), ("pad", "<u4"), ("cumulative_error_rate", "<f8"),
                           ("parent_id_hash", "<u8")])
assert NANOINFO_DTYPE.itemsize == 40


class NanoporeReadInfo:
    """_qcmodule.c:4817-4872"""
    __slots__ = ("start_time", "channel_id", "length", "cumulative_error_rate", "duration", "parent_id_hash")

    def __init__(self, rec):
        self.start_time = int(rec["start_time"])
        self.channel_id = int(rec["channel_id"])
        self.length = int(rec["length"])
        self.cumulative_error_rate = float(rec["cumulative_error_rate"])
        self.duration = float(rec["duration"])
        self.parent_id_hash = int(rec["parent_id_hash"])


class NanoStats(_DeferringArrays):
    """_qcmodule.c:4874-5430: per read start time, channel, duration, length,
    summed error rate and parent id hash, from the BAM tags when the record has any, else
    from the nanopore FASTQ header.  Reads FastqMeta.accumulated_error_rate where the
    QCMetrics pass left it (HBM), so feed an array to QCMetrics first, as the reference's
    driver does (__main__.py:279-306)."""

    def __init__(self):
        h = lib().sq_nanostats_new(context())
        if not h:
            raise MemoryError(_lib.last_error())
        self._init_defer(h)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                lib().sq_nanostats_free(self._handle)
        except Exception:
            pass

    def _run(self, arr: FastqRecordArrayView) -> None:
        # the array's QCMetrics pass, if it has one enqueued, must have left accumulated_error_rate
        # in its metas (:5314): arrays that were not staged
        wb = arr._writeback() if arr._writeback is not None else None
        if wb is not None:
            # (a view of a staging block: what that QCMetrics owes THIS block, no further -- a full drain went on into the
            # parser's open block and sealed it half full, every block of a six-module run: round 6)
            blk = getattr(arr, "_blk", None)
            blk = blk() if isinstance(blk, weakref.ReferenceType) else blk
            if blk is not None:
                wb._drain(through=blk)
            else:
                wb._drain()
        rc = lib().sq_nanostats_add_batch(self._handle, arr._device().handle)
        n = lib().sq_nanostats_last_warnings(self._handle, None, 0)
        if n:
            lengths = np.zeros(n, dtype=np.uint64)
            lib().sq_nanostats_last_warnings(self._handle, lengths.ctypes.data, n)
            for counted in lengths:  # :5247-5252
                warnings.warn("pi tag should have a valid uuid4 format with 36 characters. "
                              f"Counted {int(counted)}. Skipping tag.", UserWarning, stacklevel=2)
        check(rc)

    def add_read(self, read: FastqRecordView) -> None:
        self._add_now(FastqRecordArrayView([_require_view(read)]))

    @property
    def number_of_reads(self) -> int:
        return lib().sq_nanostats_number_of_reads(self._h)

    @property
    def minimum_time(self) -> int:
        return lib().sq_nanostats_minimum_time(self._h)

    @property
    def maximum_time(self) -> int:
        return lib().sq_nanostats_maximum_time(self._h)

    @property
    def skipped_reason(self) -> Optional[str]:
        r = lib().sq_nanostats_skipped_reason(self._h)
        return None if r is None else r.decode("ascii", "replace")

    def nano_infos(self) -> np.ndarray:
        """the NanoInfo structs of all counted reads (NANOINFO_DTYPE), one copy from HBM"""
        n = self.number_of_reads
        out = np.zeros(n, dtype=NANOINFO_DTYPE)
        if n:
            check(lib().sq_nanostats_infos(self._h, out.ctypes.data, n))
        return out

    def nano_info_iterator(self) -> Iterator[NanoporeReadInfo]:
        return (NanoporeReadInfo(rec) for rec in self.nano_infos())
