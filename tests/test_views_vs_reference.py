"""The boundary types -- FastqRecordView, FastqRecordArrayView, is_mate (_qcmodule.c:357-860) -- against the
reference's own (oracle/_ref) on arguments nobody chose: the same values back, or the same exception type and message.
CPU only (these objects live on the host); skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from tests.test_oracle_vs_reference import REF

pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/_qc.abi3.so not built (needs /root/reference)")


def outcome(fn):
    try:
        return ("ok", fn())
    except Exception as e:      # noqa: BLE001 -- the point is to compare whatever is raised
        return (type(e).__name__, str(e))


def same(got, want) -> bool:
    """equal outcomes; of a TypeError only the type (the reference's come out of PyArg_ParseTupleAndKeywords, whose wording
    of a missing or mistyped argument a Python signature cannot reproduce)"""
    if got[0] == want[0] == "TypeError":
        return True
    return got == want


def text(rng, n, alphabet):
    return "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), size=n))


ASCII = [chr(c) for c in range(32, 127)]
ODD = ASCII + ["\x7f", "\x00", "\n", "ä", "€", "\t"]


def view_of(module, args, kwargs):
    v = module.FastqRecordView(*args, **kwargs)
    return (v.name(), v.sequence(), v.qualities(), v.tags(), bytes(v.obj))


@pytest.mark.parametrize("seed", range(300))
def test_record_view(seed):
    import sequali_amd
    rng = np.random.default_rng(21000 + seed)
    L = int(rng.choice([0, 1, 4, 5, 33, 200]))
    name = text(rng, int(rng.integers(0, 30)), ODD if rng.random() < 0.2 else ASCII)
    seq = text(rng, L, ODD if rng.random() < 0.15 else list("ACGTNacgtn"))
    qual = text(rng, L if rng.random() < 0.85 else int(rng.integers(0, L + 3)), ODD if rng.random() < 0.15 else [chr(c) for c in range(33, 127)])
    args = [name, seq, qual]
    kwargs = {}
    r = rng.random()
    if r < 0.2:
        kwargs["tags"] = bytes(rng.integers(0, 256, size=int(rng.integers(0, 20))).astype(np.uint8))
    elif r < 0.25:
        kwargs["tags"] = "a str"
    elif r < 0.3:
        args[int(rng.integers(0, 3))] = rng.choice([b"bytes", 3, None])
    elif r < 0.33:
        args = args[:2]
    got, want = outcome(lambda: view_of(sequali_amd, args, kwargs)), outcome(lambda: view_of(REF, args, kwargs))
    assert same(got, want), (got, want)


def array_of(module, items):
    views = [module.FastqRecordView(*it) if isinstance(it, tuple) else it for it in items]
    arr = module.FastqRecordArrayView(views)
    out = [len(arr), bytes(arr.obj)]
    for i in list(range(len(arr))) + [-1, len(arr), -len(arr) - 1]:
        out.append(outcome(lambda i=i: (arr[i].name(), arr[i].sequence(), arr[i].qualities(), arr[i].tags())))
    return out


@pytest.mark.parametrize("seed", range(120))
def test_record_array_view(seed):
    import sequali_amd
    rng = np.random.default_rng(22000 + seed)
    items = []
    for i in range(int(rng.choice([0, 1, 2, 9, 40]))):
        L = int(rng.integers(0, 60))
        it = (text(rng, int(rng.integers(0, 20)), ASCII), text(rng, L, list("ACGTN")), text(rng, L, [chr(c) for c in range(33, 127)]))
        if rng.random() < 0.3:
            it = it + (bytes(rng.integers(0, 256, size=int(rng.integers(0, 9))).astype(np.uint8)),)
        items.append(it)
    if rng.random() < 0.15 and items:
        items[int(rng.integers(0, len(items)))] = rng.choice(["a str", 7, None])
    got, want = outcome(lambda: array_of(sequali_amd, items)), outcome(lambda: array_of(REF, items))
    assert same(got, want), (got, want)
    for bad in (None, 5, "abc"):
        assert outcome(lambda: sequali_amd.FastqRecordArrayView(bad))[0] == outcome(lambda: REF.FastqRecordArrayView(bad))[0]


def mates(module, names1, names2):
    a = module.FastqRecordArrayView([module.FastqRecordView(n, "A", "I") for n in names1])
    b = module.FastqRecordArrayView([module.FastqRecordView(n, "A", "I") for n in names2])
    return a.is_mate(b)


@pytest.mark.parametrize("seed", range(200))
def test_is_mate(seed):
    """:814-850: the ids up to the first whitespace must agree, a trailing 1 / 2 apart"""
    import sequali_amd
    rng = np.random.default_rng(23000 + seed)
    n = int(rng.choice([0, 1, 3, 8]))
    names1, names2 = [], []
    for _ in range(n):
        core = text(rng, int(rng.integers(0, 12)), list("abcXYZ09:/_."))
        kind = int(rng.integers(0, 8))
        a, b = {0: (core, core), 1: (core + "/1", core + "/2"), 2: (core + "1", core + "2"), 3: (core + " 1:N", core + " 2:N"),
                4: (core + "\tx", core + "\ty"), 5: (core + "a", core + "b"), 6: (core, core + "2"), 7: (core + "/1", core + "/1")}[kind]
        names1.append(a)
        names2.append(b)
    if rng.random() < 0.1:
        names2 = names2[:-1] if names2 else ["x"]
    got, want = outcome(lambda: mates(sequali_amd, names1, names2)), outcome(lambda: mates(REF, names1, names2))
    assert same(got, want), (got, want)
