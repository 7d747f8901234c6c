"""Shared helpers for the test-suite: golden fixture access and FASTQ text ->
(buf, metas) batches in the FastqParser layout."""
from __future__ import annotations

import glob
import json
import os
from typing import List, Tuple

import numpy as np

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_names(pattern: str) -> List[str]:
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + ".npz")))


def golden_json(name: str):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def split_fastq(text: bytes) -> Tuple[bytes, np.ndarray]:
    """Plain-python 4-line splitter: the whole text as one batch.  Offsets are
    relative to the byte after '@' (_qcmodule.c:1161-1169)."""
    metas = []
    pos, n = 0, len(text)
    while pos < n:
        assert text[pos:pos + 1] == b"@"
        e1 = text.index(b"\n", pos)
        e2 = text.index(b"\n", e1 + 1)
        e3 = text.index(b"\n", e2 + 1)
        e4 = text.index(b"\n", e3 + 1)
        name_start = pos + 1
        metas.append((name_start, e1 - name_start, e1 + 1 - name_start, e2 - (e1 + 1),
                      e3 + 1 - name_start, e4 - name_start, 0, 0.0))
        assert e4 - (e3 + 1) == e2 - (e1 + 1)
        pos = e4 + 1
    return text, np.array(metas, dtype=oracle.META_DTYPE) if metas else np.zeros(0, oracle.META_DTYPE)


def kwargs_of(g, prefix: str) -> dict:
    return json.loads(str(g[prefix + "kwargs"]))
