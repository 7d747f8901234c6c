"""ctypes binding of libsqgpu.so.  Signatures are derived from include/sqgpu.h so
that the header stays the single description of the boundary.

There is no CPU fallback: if the library is missing it is built (hipcc), and if
that fails importing the hot path fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Tuple

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "sqgpu.h")

_SCALARS = {
    "int": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "size_t": C.c_size_t,
    "double": C.c_double, "void": None, "uint32_t": C.c_uint32,
}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        base = decl.replace("const", "").replace("*", "").strip()
        if base == "char" and decl.count("*") == 1:
            return C.c_char_p
        return C.c_void_p
    base = decl.replace("const", "").strip()
    return _SCALARS[base]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{function name: (restype, [argtypes])} for every prototype in sqgpu.h"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    text = re.sub(r"typedef [^;]*;", "", text)
    text = text.replace('extern "C" {', "")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(sq_\w+)\s*\(([^;{}()]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\b\w+$", "", a).strip()  # drop the parameter name
                argtypes.append(_ctype(a))
        protos[name] = (_ctype(ret), argtypes)
    return protos


PROTOTYPES = parse_header()

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # SQ_LIB: another build of the same sources (scripts/build_asan.sh: the host side under ASan + UBSan)
        path = os.environ.get("SQ_LIB") or _build.build()
        _lib = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(_lib, name)  # AttributeError = header and library out of sync
            fn.restype = res
            fn.argtypes = args
    return _lib


def last_error() -> str:
    L = lib()
    # bytes echoed from the input -> code points, like %c; by address and length: one of them may be a zero byte
    raw = L["sq_last_error"]        # a function object of its own (CDLL.__getitem__ makes one per call): c_char_p would cut at the zero
    raw.restype = C.c_void_p
    raw.argtypes = []
    return C.string_at(raw(), L.sq_last_error_length()).decode("latin-1")


_EXC = {-1: RuntimeError, -2: ValueError, -3: MemoryError, -4: TypeError, -5: EOFError,
        -6: OverflowError, -7: RuntimeError, -8: SystemError}


def check(rc):
    """Turns a negative SQ_ERR_* code into the exception the reference raises."""
    if rc is not None and rc < 0:
        raise _EXC.get(int(rc), RuntimeError)(last_error())
    return rc


_ctx = None


def context():
    """One sq_ctx per process: device = LOCAL_RANK (one process per GPU)."""
    global _ctx
    if _ctx is None:
        dev = int(os.environ.get("SQ_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = lib().sq_init(dev)
        if not h:
            raise RuntimeError("sq_init failed: " + last_error())
        _ctx = h
    return _ctx


def synchronize() -> None:
    check(lib().sq_synchronize(context()))
