"""Synthetic FASTQ of the benchmark configurations (generator in
csrc/sq_synth_core.h).  Host and device versions produce identical bytes."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _lib
from ._lib import check, context, lib
from ._qc import META_DTYPE, FastqRecordArrayView, _DeviceBatch

DEFAULT_SEED = 20250912
ILLUMINA, ILLUMINA_R2, NANOPORE, ILLUMINA_BY_TILE, ILLUMINA_R2_BY_TILE = 0, 1, 2, 3, 4


def with_length(kind: int, length: int) -> int:
    """an Illumina kind with reads of `length` bases instead of 150 (SQ_SYNTH_KIND_LEN)"""
    return kind | (int(length) << 8)

ILLUMINA_PROBES = ("AGATCGGAAGAG", "TGGAATTCTCGG", "GATCGTCGGACT", "CTGTCTCTTATA",
                   "GGGGGGGGGGGG", "AAAAAAAAAAAA")  # adapters/adapter_list.tsv:8-15
NANOPORE_PROBES = ("TTACGTATTGCT", "GCAATACGTAAC", "CTTGCGGGCGGC", "GGTAGTAGGTTC",
                   "GAGGCGAGCGGT", "CAAGATACGCAC", "GTGACTTGCCTG", "ATCGCCTACCGT",
                   "TCTATCTTCTTT", "TCTTCAGAGGAG", "GATATTGCTGGG", "TGATATTGCTTT",
                   "GTACGTATTGCT", "ACGTAACTGAAC")  # adapters/adapter_list.tsv:35-57


def host_records(kind: int, first: int, n: int, seed: int = DEFAULT_SEED) -> Tuple[bytes, np.ndarray]:
    """(FASTQ text, metas) of records [first, first+n), generated on the host."""
    size = lib().sq_synth_bytes(kind, seed, first, n)
    buf = np.zeros(size, dtype=np.uint8)
    metas = np.zeros(n, dtype=META_DTYPE)
    check(lib().sq_synth_host(kind, seed, first, n, buf.ctypes.data, size, metas.ctypes.data))
    return buf.tobytes(), metas


def host_array(kind: int, first: int, n: int, seed: int = DEFAULT_SEED) -> FastqRecordArrayView:
    buf, metas = host_records(kind, first, n, seed)
    return FastqRecordArrayView._from_buffer(buf, metas)


def device_array(kind: int, first: int, n: int, seed: int = DEFAULT_SEED) -> FastqRecordArrayView:
    """Records generated straight into HBM by a kernel (never touch the host)."""
    h = lib().sq_synth_device(context(), kind, seed, first, n)
    if not h:
        raise MemoryError(_lib.last_error())
    return FastqRecordArrayView._from_device(_DeviceBatch(h))


def illumina_fastq(first: int, n: int, seed: int = DEFAULT_SEED) -> bytes:
    return host_records(ILLUMINA, first, n, seed)[0]


def illumina_paired_fastq(first: int, n: int, seed: int = DEFAULT_SEED) -> Tuple[bytes, bytes]:
    return host_records(ILLUMINA, first, n, seed)[0], host_records(ILLUMINA_R2, first, n, seed)[0]


def nanopore_fastq(first: int, n: int, seed: int = DEFAULT_SEED) -> bytes:
    return host_records(NANOPORE, first, n, seed)[0]
