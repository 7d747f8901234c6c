"""NanoStats (_qcmodule.c:4804-5430) against tests/golden/nanostats_cases.npz: for every
case the arrays the reference's NanoStats was fed (bytes + FastqMeta structs carrying
accumulated_error_rate from its QCMetrics) and what it reported -- NanoInfo of every read,
number_of_reads, minimum / maximum time, skipped_reason, the exception and the warnings.
CPU: the oracle.  GPU: sequali_amd.NanoStats through libsqgpu.so."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import GOLDEN

CASES = np.load(os.path.join(GOLDEN, "nanostats_cases.npz"))
NAMES = [str(n) for n in CASES["names"]]
FIELDS = ("start_time", "duration", "channel_id", "length", "cumulative_error_rate", "parent_id_hash")

ORACLE_ERRORS = {1: ("ValueError", "truncated tags"), 2: ("ValueError", "Invalid type for array {0}"),
                 3: ("ValueError", "Unknown tag type {0}"),
                 4: ("RuntimeError", "Wrong tag type for '{0}{1}' expected '{3}' got '{2}'"), 5: ("SystemError", None)}


def arrays_of(name):
    for k in range(int(CASES[name + "_n_arrays"])):
        obj = CASES[f"{name}_obj{k}"].tobytes()
        metas = np.frombuffer(CASES[f"{name}_metas{k}"].tobytes(), dtype=oracle.META_DTYPE).copy()
        yield obj, metas


def expected(name):
    res = json.loads(str(CASES[name + "_result"]))
    n, tmin, tmax = (int(x) for x in CASES[name + "_scalars"])
    return res, n, tmin, tmax, CASES[name + "_infos"]


def check_infos(got, want):
    assert len(got) == len(want)
    for f in FIELDS:
        a, b = got[f], want[f]
        if a.dtype.kind == "f":
            a, b = a.view(f"u{a.dtype.itemsize}"), b.view(f"u{b.dtype.itemsize}")
        assert np.array_equal(a, b), f


@pytest.mark.parametrize("name", NAMES)
def test_oracle_nanostats_matches_reference(name):
    res, n, tmin, tmax, infos = expected(name)
    ns = oracle.NanoStats()
    error = None
    try:
        for obj, metas in arrays_of(name):
            ns.add(obj, metas)
    except oracle.NanoStatsError as e:
        kind, msg = ORACLE_ERRORS[e.code]
        c = e.chars.decode("latin-1")
        expect = {"st": "Z", "du": "f", "pi": "Z"}.get(c[:2], "?")
        error = (kind, msg.format(c[0], c[1], c[2], expect) if msg else None)
    if "error" in res:
        assert error is not None and error[0] == res["error"]
        if error[1] is not None:
            assert error[1] == res["message"]
    else:
        assert error is None
    assert ns.number_of_reads == n
    assert (ns.minimum_time, ns.maximum_time) == (tmin, tmax)
    assert ns.skipped == (res["skipped_reason"] is not None)
    assert ns.pi_warnings == len(res["warnings"])
    check_infos(ns.nano_infos(), infos)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_nanostats_matches_reference(name):
    from sequali_amd import FastqRecordArrayView, NanoStats
    res, n, tmin, tmax, infos = expected(name)
    ns = NanoStats()
    error = None
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        try:
            for obj, metas in arrays_of(name):
                ns.add_record_array(FastqRecordArrayView._from_buffer(obj, metas))
            ns.flush()   # small arrays are staged: their exceptions and warnings come out here
        except (ValueError, RuntimeError, SystemError) as e:
            error = (type(e).__name__, str(e))
    if "error" in res:
        assert error is not None and error[0] == res["error"]
        if res["error"] != "SystemError":
            assert error[1] == res["message"]
    else:
        assert error is None
    assert ns.number_of_reads == n
    assert (ns.minimum_time, ns.maximum_time) == (tmin, tmax)
    assert ns.skipped_reason == res["skipped_reason"]
    assert [str(w.message) for w in caught] == res["warnings"]
    got = list(ns.nano_info_iterator())
    assert len(got) == len(infos)
    for g, w in zip(got, infos):
        assert g.start_time == int(w["start_time"]) and g.channel_id == int(w["channel_id"])
        assert g.length == int(w["length"]) and g.parent_id_hash == int(w["parent_id_hash"])
        assert np.float32(g.duration).view(np.uint32) == w["duration"].view(np.uint32)
        assert np.float64(g.cumulative_error_rate).view(np.uint64) == w["cumulative_error_rate"].view(np.uint64)
