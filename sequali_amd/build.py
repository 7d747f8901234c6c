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
SOURCES = ["sq_span.hip", "sq_span_w6.hip", "sq_pair.hip", "sq_qc.hip", "sq_api.hip", "sq_ends.hip", "sq_nano.hip", "sq_feed.hip", "sq_dist.hip", "sq_hostsimd.cpp"]   # the slowest first
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Xclang", "-no-enable-noundef-analysis"]


# Builds of k_span (sq_span_kernel.h: <NW, AD, SEG, W4T, SPLIT, LONG, PT, PAIR>) that the DEFAULT dispatch launches (sq_span_launch,
# sq_span_launch_sorted, sq_span_launch_long; tests/test_gpu_routes.py asserts the same routes on the GPU).  A register
# spilled inside their loop is reloaded behind an `s_waitcnt vmcnt(0)`, which also waits for the span in flight: the
# dispatcher would fall back to another kernel without a word (span_build_spills; round 2 lost a third of a route's
# speed that way), so the build fails instead.  The run-time guard stays for experiment builds.
def default_route_builds():
    b = lambda x: "Lb1" if x else "Lb0"   # noqa: E731
    out = []
    for nw in range(1, 9):
        out.append((nw, False, False, nw >= 6, False))                 # QCMetrics alone, one read length: one wave for both streams, from 6 windows on a wave per stream (round 5)
        out.append((nw, True, False, True, False))                     # + AdapterCounter: a wave per stream
        # reads of many lengths: one wave for both streams where that build exists, else a wave per stream
        out.append((nw, False, True, nw == 6, False))
        out.append((nw, True, True, nw >= 6, False))
    for ad in (False, True):
        out.append((8, ad, True, True, True))                          # segments of long reads
    names = ["k_spanILi%dE%sE%sELi3E%sE%sELb0ELi0EE" % (nw, b(ad), b(seg), b(split), b(lng)) for nw, ad, seg, split, lng in out]
    # QCMetrics with PerTileQuality riding along; with the ends of read 2 written / the overlap scan on read 1 (sq_pair.hip)
    names += ["k_spanILi%dELb0ELb0ELi3ELb0ELb0ELb1ELi%dEE" % (nw, pair) for nw in range(1, 9) for pair in (0, 1, 2)]
    names += ["k_ptspanILi%dEE" % nw for nw in range(1, 9)] + ["k_isz_spanILi%dEE" % nw for nw in range(1, 9)]
    # adapters of 14-25 characters (sq_span_w6.hip; the default since round 5): batches of one read length from 5 windows on (a wave
    # per stream), every window count the length-sorted route can meet (one wave for both streams up to 5 windows, a wave per stream beyond)
    names += ["k_spanILi%dELb1ELb0ELi6ELb1ELb0ELb0ELi0EE" % nw for nw in range(5, 9)]
    names += ["k_spanILi%dELb1ELb1ELi6E%sELb0ELb0ELi0EE" % (nw, b(nw >= 6)) for nw in range(3, 9)]
    # batches of one read length of 225-256 bases with adapters, and adapters of 14-25 characters below 129 bases: the round-1 kernel
    names += ["6k_wideILb1EE"]
    return names


def parse_resource_remarks(stderr: str):
    """{mangled kernel name: {"vgprs": .., "scratch": .., "spill": ..}} from -Rpass-analysis=kernel-resource-usage"""
    import re
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[{"VGPRs": "vgprs", "VGPRs Spill": "spill", "SGPRs Spill": "sgpr_spill", "ScratchSize [bytes/lane]": "scratch",
                 "Occupancy [waves/SIMD]": "occupancy"}[m.group(1)]] = int(m.group(2))
    return out


def check_default_routes(resources) -> None:
    missing, spilling = [], []
    for want in default_route_builds():
        hit = [(k, v) for k, v in resources.items() if want in k]
        if not hit:
            missing.append(want)
        for k, v in hit:
            if v.get("spill", 0) or v.get("scratch", 0):
                spilling.append(f"{k}: {v}")
    if missing or spilling:
        raise RuntimeError("kernels of a default route are missing from the build or use scratch memory:\n" +
                           "\n".join(missing + spilling))


# The f64 chains of k_span end in steps that exist for some read lengths only: scalar compares of U against constants guard them.
# Round 5 met a build in which hipcc (ROCm 7.2.0, clang 22) had DROPPED three of them after reading a fact about U out of an
# unrelated expression (every read of 1-15 and 33-47 bases summed the text behind its qualities); the sources now keep
# U opaque there.  tests/test_gpu_vs_oracle.py sweeps every length on a GPU; this is the same question asked of the listing, where no
# GPU is: the builds for batches of one read length of up to 64 bases must hold at least the compares the good build of this
# compiler holds.  Another compiler: no verdict (its code may be right with other instructions).
CHAIN_GUARDS = {"version": "7.2.", "min_scalar_compares": {1: 11, 2: 11}}   # k_span<NW, ., uniform>: s_cmp_gt_u32 + s_cmp_lt_u32


def check_chain_guards(obj: str, hipcc: str) -> None:
    import re
    import tempfile
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    if "HIP version: " + CHAIN_GUARDS["version"] not in ver:
        print("build.py: hipcc is not the ROCm %sx the chain-guard check knows: skipped" % CHAIN_GUARDS["version"], file=sys.stderr)
        return
    llvm = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat"), os.path.join(tmp, "dev.co")
        r = subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True, text=True)
        if r.returncode == 0:
            r = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], capture_output=True, text=True)
        if r.returncode != 0:
            print("build.py: the device code of %s could not be taken out: chain-guard check skipped\n%s" % (obj, r.stderr), file=sys.stderr)
            return
        listing = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", dev], capture_output=True, text=True).stdout
    bad = []
    for body in re.split(r"\n(?=[0-9a-f]{16} <)", listing):
        m = re.match(r"[0-9a-f]{16} <(\S*k_spanILi([12])ELb[01]ELb0ELi3E\S*)>", body)
        if not m:
            continue
        n = body.count("s_cmp_gt_u32") + body.count("s_cmp_lt_u32")
        if n < CHAIN_GUARDS["min_scalar_compares"][int(m.group(2))]:
            bad.append("%s: %d scalar compares" % (m.group(1), n))
    if bad:
        raise RuntimeError("k_span builds for short reads have lost guards of their f64 chains (see build.py::CHAIN_GUARDS):\n" + "\n".join(bad))


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libsqgpu.so cannot be built")
    return exe


def _sha(paths, extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for path in paths:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(HERE, "..", "include", "sqgpu.h")]


def source_key(src: str) -> str:
    """what an object file was made from: its source, every header, the flags.  Content, not mtimes: a library that
    travelled with a snapshot of the tree (gpurun) or was checked out beside newer sources must not pass as current."""
    return _sha([os.path.join(CSRC, src)] + _headers(), " ".join(FLAGS))


def library_key() -> str:
    return _sha([os.path.join(CSRC, src) for src in SOURCES] + _headers(), " ".join(FLAGS))


def _read(path: str) -> str:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


KEYFILE = LIB + ".key"


def _stale() -> bool:
    return not os.path.exists(LIB) or _read(KEYFILE) != library_key()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles and links under a file lock, into temporary names that are renamed into place:
    ranks of one torchrun job that all find the library stale neither compile into the same
    object files at once nor dlopen a half-written library.  A source is compiled again when the
    hash of what it is made from differs from the one recorded beside its object file."""
    if not force and not _stale():
        return LIB
    import fcntl
    import json
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():     # another process built it while this one waited
            return LIB
        tag = f".{os.getpid()}.tmp"

        def compile_one(src: str):
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            key, resfile = source_key(src), obj + ".res.json"
            if not force and os.path.exists(obj) and _read(obj + ".key") == key and os.path.exists(resfile):
                with open(resfile) as f:
                    return obj, json.load(f)
            flags = FLAGS + ["-Rpass-analysis=kernel-resource-usage"] if src.endswith(".hip") else \
                ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall"]   # .cpp: host only
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj + tag]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
            res = parse_resource_remarks(r.stderr)
            if verbose and r.stderr:
                print("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and l.strip() not in ("^", "")
                                and not l.lstrip().split("|")[0].strip().isdigit()), file=sys.stderr)
            os.replace(obj + tag, obj)
            with open(resfile, "w") as f:
                json.dump(res, f)
            with open(obj + ".key", "w") as f:
                f.write(key)
            return obj, res

        with ThreadPoolExecutor(max_workers=8) as pool:
            done = list(pool.map(compile_one, SOURCES))
        objs = [o for o, _ in done]
        resources = {}
        for _, res in done:
            resources.update(res)
        with open(os.path.join(objdir, "resources.json"), "w") as f:
            json.dump(resources, f, indent=0, sort_keys=True)
        check_default_routes(resources)   # before the link: a library with a spilling default route is not made
        check_chain_guards(os.path.join(objdir, "sq_span.o"), hipcc)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + tag, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        os.replace(LIB + tag, LIB)
        with open(KEYFILE, "w") as f:
            f.write(library_key())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
