"""Builds libsqgpu.so (hand-written HIP for gfx950 + the C ABI of include/sqgpu.h)
in-tree with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsqgpu.so")
SOURCES = ["sq_api.hip", "sq_qc.hip", "sq_span.hip", "sq_ends.hip", "sq_nano.hip", "sq_feed.hip", "sq_hostsimd.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libsqgpu.so cannot be built")
    return exe


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "sqgpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles and links under a file lock, into temporary names that are renamed into place:
    ranks of one torchrun job that all find the library stale neither compile into the same
    object files at once nor dlopen a half-written library."""
    if not force and not _stale():
        return LIB
    import fcntl
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():     # another process built it while this one waited
            return LIB
        tag = f".{os.getpid()}.tmp"

        def compile_one(src: str) -> str:
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            flags = FLAGS if src.endswith(".hip") else ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall"]   # .cpp: host only
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj + tag]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
            if verbose and r.stderr:
                print(r.stderr, file=sys.stderr)
            os.replace(obj + tag, obj)
            return obj

        with ThreadPoolExecutor(max_workers=7) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + tag, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        os.replace(LIB + tag, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
