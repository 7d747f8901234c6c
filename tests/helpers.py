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
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + ".npz")))
    if not names:   # a parametrisation that silently expands to nothing tests nothing
        raise FileNotFoundError(f"no golden fixture matches {pattern!r} in {GOLDEN}")
    return names


def golden_text(g, key: str) -> bytes:
    """Input text of a fixture.  The synth_* fixtures keep (kind, first, n, seed) of the build's
    counter-based generator and the SHA-256 of the bytes the reference was run on instead of the
    bytes: regenerate (host generator of libsqgpu.so, no GPU needed) and check."""
    if key in g.files:
        return g[key].tobytes()
    import hashlib
    from sequali_amd import synth
    kind, first, n, seed = (int(x) for x in g[key + "_gen"])
    text = synth.host_records(kind, first, n, seed)[0]
    if hashlib.sha256(text).digest() != g[key + "_sha256"].tobytes():
        raise AssertionError(f"the synthetic generator no longer produces the bytes fixture {key} was captured on")
    return text


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


def with_env(env: dict, fn):
    """Runs fn with the SQ_* switches of `env` in force: the library reads them once, so it is
    told to read them again (sq_knobs_reload) after every change."""
    from sequali_amd._lib import lib
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    lib().sq_knobs_reload()
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib().sq_knobs_reload()
