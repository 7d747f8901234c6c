#!/usr/bin/env python3
"""bench.py -- Gbases/s of the per-read QC hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], per GPU): 100 M x 150 bp synthetic
single-end Illumina FASTQ resident in HBM, QCMetrics + AdapterCounter (the six
Illumina probes) in one fused pass.  A step is one pass over all records of
the rank's shard, followed (N > 1) by the RCCL all-reduce of the count tables.
Scaling is weak: every rank holds its own 100 M records (records
[rank * R, (rank + 1) * R) of the counter-based generator).

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline     -- fused-pass kernel: algorithmic bytes / launch duration (HIP
                  events on the library's stream) against the 8 TB/s HBM peak
  cpu_baseline -- the reference's own C (oracle/_ref, built from its sources)
                  or the C port (oracle/), one thread, on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

READ_LEN = 150
# 2 B/base + 40 B/read meta + 8 B/read accumulated_error_rate write-back
ALGO_BYTES_PER_READ = 2 * READ_LEN + 40 + 8
HBM_PEAK_GBPS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None,
                    help="records per GPU (default 100 M illumina / 1 M nanopore)")
    ap.add_argument("--batch-reads", type=int, default=None,
                    help="records per launch (default 25 M illumina / 1 M nanopore)")
    ap.add_argument("--cpu-passes", type=int, default=3,
                    help="times the CPU baseline walks its sample (about 12 s of CPU work by default)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="records of the CPU baseline sample (0 = skip)")
    ap.add_argument("--configs", default=None,
                    help="comma-separated entries of other_configs to run (default: all); for profiling one configuration at a time")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip configs 3, 4 and the ragged variant (other_configs of the JSON line)")
    ap.add_argument("--modules", default="qc,adapter", help="qc,adapter[,pertile]")
    ap.add_argument("--kind", default="illumina", choices=["illumina", "nanopore"],
                    help="illumina: 150 bp (configs 2/5); nanopore: ~10 kb variable length (config 4)")
    args = ap.parse_args()
    if args.reads is None:
        args.reads = 100_000_000 if args.kind == "illumina" else 1_000_000
    if args.batch_reads is None:
        args.batch_reads = 25_000_000 if args.kind == "illumina" else 1_000_000
    return args


class HipEvents:
    """hipEvent timing on a given stream through libamdhip64 (torch.cuda.Event only
    sees torch's own stream)."""

    def __init__(self, stream):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.stream = ctypes.c_void_p(stream)
        self.pairs = []

    def _new(self):
        ev = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(ev)) == 0
        return ev

    def start(self):
        a, b = self._new(), self._new()
        assert self.hip.hipEventRecord(a, self.stream) == 0
        self.pairs.append((a, b))

    def stop(self):
        assert self.hip.hipEventRecord(self.pairs[-1][1], self.stream) == 0

    def durations_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            assert self.hip.hipEventSynchronize(b) == 0
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            out.append(ms.value)
            self.hip.hipEventDestroy(a)
            self.hip.hipEventDestroy(b)
        self.pairs = []
        return out


def dominant_kernel(kind, mods):
    """the kernel sq_fused_add_batch launches for the synthetic batch of this run"""
    if kind != "illumina":
        return "k_span<LONG> (long reads: k_read_sums + k_span<8,AD,LONG> + k_long_gc_bins + k_long_ea + k_adapter_first)"
    if "pertile" in mods:
        return "k_pass (fused per-base pass, tile-sorted order)"
    if "adapter" in mods and "qc" in mods:
        import os
        if os.environ.get("SQ_SPAN", "1") != "0":
            return "k_span<AD> (fused per-base pass, one read length, records streamed through LDS)"
        return "k_wide<AD> (fused per-base pass, one read length)"
    if "qc" in mods:
        import os
        if os.environ.get("SQ_SPAN", "1") != "0":
            return "k_span (QCMetrics, one read length, records streamed through LDS)"
        return "k_ring (QCMetrics, one read length)"
    return "k_pass (fused per-base pass)"


def compact_route(route: str) -> str:
    """"a+b+a+b" -> "2 x (a+b)": the route of a step that walks several batches"""
    parts = route.split("+") if route else []
    for period in range(1, len(parts) // 2 + 1):
        if len(parts) % period == 0 and parts == parts[:period] * (len(parts) // period):
            return f"{len(parts) // period} x ({'+'.join(parts[:period])})"
    return "+".join(parts)


def host_cpu():
    """model name and core count of the box the CPU baseline ran on"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count()


def csrc_sha():
    """SHA-256 over the kernel sources: profiles/traffic.json records the one it was measured on (every file of csrc/ but the
    two that hold no device code: the parser's host side, sq_feed.hip, and its byte scans, sq_hostsimd.cpp)"""
    import hashlib
    from sequali_amd import build as _build
    h = hashlib.sha256()
    h.update(" ".join(_build.FLAGS).encode())     # the same sources under other compiler flags are other kernels
    d = os.path.join(ROOT, "sequali_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name in ("sq_feed.hip", "sq_hostsimd.cpp"):
            continue
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(sample_reads: int, passes: int = 1):
    """One QC thread (the reference's second thread only decompresses,
    __main__.py:189-192) over the first `sample_reads` records of the workload."""
    import numpy as np
    from sequali_amd import synth
    dev = synth.device_array(synth.ILLUMINA, 0, sample_reads)
    buf, metas = dev._batch.download()
    del dev
    bases = int(metas["sequence_length"].sum())
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    kind = "port"
    if os.path.exists(os.path.join(ref_dir, "_qc.abi3.so")):
        try:
            sys.path.insert(0, ref_dir)
            import _qc as ref  # the reference's own extension, compiled from its sources
            kind = "reference"
        except Exception:
            kind = "port"
    if kind == "reference":
        arrays = list(ref.FastqParser(io.BytesIO(buf.tobytes())))  # parsing is not timed
        m, a = ref.QCMetrics(), ref.AdapterCounter(list(synth.ILLUMINA_PROBES))
        t0 = time.perf_counter()
        for _ in range(passes):
            for arr in arrays:
                m.add_record_array(arr)
                a.add_record_array(arr)
        m.base_count_table()
        dt = time.perf_counter() - t0
        check = int(np.array(m.base_count_table(), dtype=np.uint64).sum())
    else:
        from oracle import oracle
        m, a = oracle.QCMetrics(), oracle.AdapterCounter(list(synth.ILLUMINA_PROBES))
        t0 = time.perf_counter()
        for _ in range(passes):
            m.add(buf, metas)
            a.add(buf, metas)
        dt = time.perf_counter() - t0
        check = int(m.base_count_table().sum())
    assert check == bases * passes
    model, cores = host_cpu()
    return {"value": round(bases * passes / dt / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": kind,
            "host_cpu": model, "host_cores": cores,
            "sample": f"first {sample_reads} records of the workload ({bases} bases) x {passes} passes, "
                      f"QCMetrics+AdapterCounter on one thread (the reference's second thread only decompresses), {dt:.2f} s"}


def _reference_module():
    """the reference's own extension (oracle/_ref, compiled from its sources by oracle/Makefile), or None"""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(ref_dir, "_qc.abi3.so")):
        return None
    try:
        if ref_dir not in sys.path:
            sys.path.insert(0, ref_dir)
        import _qc as ref
        return ref
    except Exception:
        return None


def cpu_baselines_other(only=None):
    """The reference's C on one thread beside configs 3 and 4 and the six-module loop: bounded samples of the same
    synthetic records (host generator: identical bytes), parsed by the reference's own FastqParser (not timed), fed
    array by array as __main__.py:279-306 does."""
    from sequali_amd import synth
    ref = _reference_module()
    if ref is None:
        return {}
    model, cores = host_cpu()
    out = {}

    def arrays_of(kind, n):
        text, metas = synth.host_records(kind, 0, n)
        return list(ref.FastqParser(io.BytesIO(text))), int(metas["sequence_length"].sum())

    def entry(bases, dt, sample):
        return {"value": round(bases / dt / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "reference",
                "host_cpu": model, "host_cores": cores, "sample": sample + f", {dt:.2f} s"}

    if only is None or "config3_paired" in only:
        n = 1_000_000
        (text1, metas1), (text2, metas2) = synth.host_records(synth.ILLUMINA, 0, n), synth.host_records(synth.ILLUMINA_R2, 0, n)
        b1, b2 = int(metas1["sequence_length"].sum()), int(metas2["sequence_length"].sum())
        p1, p2 = ref.FastqParser(io.BytesIO(text1)), ref.FastqParser(io.BytesIO(text2))
        pairs = [(x, p2.read(len(x))) for x in p1]      # the mates' arrays cut at the same record counts, as the driver does
        m1, m2, t1, t2, z = ref.QCMetrics(), ref.QCMetrics(), ref.PerTileQuality(), ref.PerTileQuality(), ref.InsertSizeMetrics()
        t0 = time.perf_counter()
        for x, y in pairs:
            m1.add_record_array(x); t1.add_record_array(x)
            m2.add_record_array(y); t2.add_record_array(y)
            z.add_record_array_pair(x, y)
        dt = time.perf_counter() - t0
        assert z.total_reads == n and m1.number_of_reads == n
        c3 = entry(b1 + b2, dt, f"first {n} pairs of the workload ({b1 + b2} bases), (QCMetrics + PerTileQuality) x 2 + "
                   "InsertSizeMetrics on one thread")
        for name in ("config3_paired", "config3_paired_by_tile", "config3_paired_five_calls_unfused"):
            out[name] = c3
        del pairs
    if only is None or "config4_nanopore" in only:
        n = 100_000
        arrays, bases = arrays_of(synth.NANOPORE, n)
        m, a = ref.QCMetrics(), ref.AdapterCounter(list(synth.NANOPORE_PROBES))
        t0 = time.perf_counter()
        for x in arrays:
            m.add_record_array(x)
            a.add_record_array(x)
        dt = time.perf_counter() - t0
        assert m.number_of_reads == n
        out["config4_nanopore"] = entry(bases, dt, f"first {n} reads of the workload ({bases} bases), QCMetrics + AdapterCounter "
                                        "(14 probes) on one thread")
        del arrays
    if only is None or "single_end_six_modules" in only:
        n = 2_000_000
        arrays, bases = arrays_of(synth.ILLUMINA, n)
        mods = (ref.QCMetrics(), ref.AdapterCounter(list(synth.ILLUMINA_PROBES)), ref.PerTileQuality(),
                ref.OverrepresentedSequences(), ref.NanoStats(),
                ref.DedupEstimator(front_sequence_offset=64, back_sequence_offset=0))
        t0 = time.perf_counter()
        for x in arrays:
            for mod in mods:
                mod.add_record_array(x)
        dt = time.perf_counter() - t0
        assert mods[0].number_of_reads == n
        out["single_end_six_modules"] = entry(bases, dt, f"first {n} reads of the workload ({bases} bases), the six single-end modules of "
                                              "__main__.py:279-306 on one thread")
        per = {}
        for label, make in (("overrep_alone", lambda: ref.OverrepresentedSequences()),
                            ("dedup_single_end", lambda: ref.DedupEstimator(front_sequence_offset=64, back_sequence_offset=0))):
            mod = make()
            t0 = time.perf_counter()
            for x in arrays:
                mod.add_record_array(x)
            dt = time.perf_counter() - t0
            per[label] = entry(bases, dt, f"first {n} reads of the workload ({bases} bases), the module alone on one thread")
        out.update(per)
    return out


def other_configs(lib, ctx, steps, warmup, only=None):
    """BASELINE configs 3 and 4 and the ragged variant of config 2, each timed like the headline:
    records resident in HBM, `steps` passes behind `warmup`, HIP events on the library's stream
    around every pass; achieved = algorithmic bytes / pass time."""
    import numpy as np
    from sequali_amd import (AdapterCounter, FusedPass, InsertSizeMetrics, PerTileQuality, QCMetrics, _lib, synth)

    def run(name, workload, kernel, arrays_bases_reads, make, step, check):
        bases, reads, algo = arrays_bases_reads
        objs = make()
        lib.sq_route_reset(ctx)
        for _ in range(max(warmup, 1)):
            step(objs)
        _lib.synchronize()
        route = route_of_step()     # the kernels one step launched (sq_last_route): a silent fallback would show here
        ev = HipEvents(lib.sq_stream_handle(ctx))
        t0 = time.perf_counter()
        for _ in range(steps):
            ev.start()
            step(objs)
            ev.stop()
        _lib.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms = ev.durations_ms()
        avg = sum(ms) / len(ms)
        achieved = algo / (avg * 1e-3) / 1e9
        return {"workload": workload, "value": round(bases / dt / 1e9, 3), "unit": "Gbases/s",
                "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": max(warmup, 1), "route": route,
                "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBPS, 5), "kernel": kernel,
                             "avg_launch_ms": round(avg, 4), "algorithmic_bytes_per_step": int(algo)},
                "checks": check(objs, steps + warmup)}

    def clear(f):
        f.qc_metrics._pending.clear()

    def route_of_step():
        return compact_route((lib.sq_last_route(ctx) or b"").decode())

    def wanted(name):   # --configs: a subset of the entries (profiling one configuration at a time)
        return only is None or name in only

    out = {}
    # ---- batches of one read length other than 150: 200 and 250 bases (2 x 250 is a real Illumina length) ----
    for L in (200, 250):
        if not wanted(f"uniform_{L}bp"):
            continue
        n, per = 50_000_000, 25_000_000
        batches = [synth.device_array(synth.with_length(synth.ILLUMINA, L), k * per, per) for k in range(n // per)]
        bases = sum(b._batch.total_bases for b in batches)

        def uniform_step(f, batches=batches):
            for b in batches:
                f.add_record_array(b)
                clear(f)

        out[f"uniform_{L}bp"] = run(
            f"uniform{L}", f"{n} x {L} bp synthetic single-end reads, QCMetrics + AdapterCounter fused, records resident in HBM",
            "the kernel `route` names", (bases, n, 2 * bases + 48 * n),
            lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))), uniform_step,
            lambda f, passes, bases=bases, n=n: {
                "base_table_sum_ok": bool(int(np.array(f.qc_metrics.base_count_table(), np.uint64).sum()) == bases * passes),
                "phred_scores_sum_ok": bool(int(np.array(f.qc_metrics.phred_scores(), np.uint64).sum()) == n * passes)})
        del batches
    # ---- ragged: the config-2 records cut to 50 .. 150 bases (what adapter trimming leaves) ----
    if wanted("ragged_50_150"):
        n, per = 50_000_000, 25_000_000
        batches = [synth.device_array(synth.ILLUMINA, k * per, per) for k in range(n // per)]
        for k, b in enumerate(batches):
            _lib.check(lib.sq_synth_trim(b._batch.handle, 77 + k, 50))
        bases = sum(b._batch.total_bases for b in batches)

        def ragged_step(f):
            for b in batches:
                f.add_record_array(b)
                clear(f)

        out["ragged_50_150"] = run(
            "ragged", f"{n} synthetic reads of 50..150 bases (the 150 bp records cut by a hash of the record index), "
            "QCMetrics + AdapterCounter fused, records resident in HBM",
            "k_span<NW,AD,SEG> x 4 window counts (rows in order of length, spans of 16 reads of one length; k_span_scatter in front: the batch knows how many reads have each length)", (bases, n, 2 * bases + 48 * n),
            lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))), ragged_step,
            lambda f, passes: {"base_table_sum_ok": bool(int(np.array(f.qc_metrics.base_count_table(), np.uint64).sum()) == bases * passes),
                               "phred_scores_sum_ok": bool(int(np.array(f.qc_metrics.phred_scores(), np.uint64).sum()) == n * passes)})
        del batches
    # ---- config 3: 100 M pairs, (QCMetrics + PerTileQuality) x 2 + InsertSizeMetrics ----
    # three entries: the reads with a random tile each (SURVEY 8d's generator: the config BASELINE.json names), the same pairs in the
    # order a sequencer writes them (65536 reads of a tile in a row) -- both through PairedPass, what the driver calls per pair of
    # arrays (sq_paired_add_batches: PerTileQuality's tile ids and the overlap scan ride in QCMetrics' passes, csrc/sq_pair.hip;
    # reads of one tile in a row also keep PerTileQuality's sums in the pass) -- and the random tiles through the five separate
    # calls of the reference's driver loop with SQ_PT_FUSED=0 (the seven passes of rounds 1-3, for comparison)
    from sequali_amd import PairedPass
    n, per = 100_000_000, 25_000_000
    name_len = 36
    for entry, kinds, fused in (("config3_paired", (synth.ILLUMINA, synth.ILLUMINA_R2), True),
                                ("config3_paired_by_tile", (synth.ILLUMINA_BY_TILE, synth.ILLUMINA_R2_BY_TILE), True),
                                ("config3_paired_five_calls_unfused", (synth.ILLUMINA, synth.ILLUMINA_R2), False)):
        if not wanted(entry):
            continue
        r1 = [synth.device_array(kinds[0], k * per, per) for k in range(n // per)]
        r2 = [synth.device_array(kinds[1], k * per, per) for k in range(n // per)]
        bases = sum(b._batch.total_bases for b in r1) + sum(b._batch.total_bases for b in r2)

        def c3_make():
            return (FusedPass(QCMetrics(), None, PerTileQuality()), FusedPass(QCMetrics(), None, PerTileQuality()), InsertSizeMetrics())

        def c3_step(o, r1=r1, r2=r2):
            fa, fb, isz = o
            for a, b in zip(r1, r2):
                fa.add_record_array(a); clear(fa)
                fb.add_record_array(b); clear(fb)
                isz.add_record_array_pair(a, b)

        def c3_step_fused(o, r1=r1, r2=r2):
            fa, fb, isz = o
            pp = PairedPass(fa.qc_metrics, fa.per_tile_quality, fb.qc_metrics, fb.per_tile_quality, isz)
            for a, b in zip(r1, r2):
                pp.add_record_array_pair(a, b)
                clear(fa); clear(fb)

        def c3_check(o, passes, bases=bases):
            fa, fb, isz = o
            return {"base_table_sums_ok": bool(int(np.array(fa.qc_metrics.base_count_table(), np.uint64).sum()) +
                                               int(np.array(fb.qc_metrics.base_count_table(), np.uint64).sum()) == bases * passes),
                    "pertile_reads_ok": bool(fa.per_tile_quality.number_of_reads == n * passes and fb.per_tile_quality.number_of_reads == n * passes),
                    "insert_size_pairs_ok": bool(isz.total_reads == n * passes)}

        if not fused:
            os.environ["SQ_PT_FUSED"] = "0"
            lib.sq_knobs_reload()
        try:
            out[entry] = run(
                "config3", f"{n} x 150 bp synthetic pairs ({'reads tile by tile, 65536 in a row' if kinds[0] != synth.ILLUMINA else 'a random tile per read'}), "
                "(QCMetrics + PerTileQuality) x 2 + InsertSizeMetrics, records resident in HBM" + (", through PairedPass" if fused else ", five calls per pair of arrays, SQ_PT_FUSED=0"),
                "the kernels `route` names", (bases, 2 * n, 2 * bases + (48 + name_len) * 2 * n),
                c3_make, c3_step_fused if fused else c3_step, c3_check)
        finally:
            if not fused:
                os.environ.pop("SQ_PT_FUSED", None)
                lib.sq_knobs_reload()
        del r1, r2
    # ---- config 4: 1 M x ~10 kb nanopore reads, QCMetrics + AdapterCounter (14 probes) ----
    if wanted("config4_nanopore"):
        n = 1_000_000
        arr = synth.device_array(synth.NANOPORE, 0, n)
        bases = arr._batch.total_bases

        def c4_step(f):
            f.add_record_array(arr)
            clear(f)

        out["config4_nanopore"] = run(
            "config4", f"{n} synthetic nanopore reads (~10 kb, 200 .. 100000), QCMetrics + AdapterCounter (14 probes), records resident in HBM",
            "k_span<8,AD,LONG> (segments of 256 positions of the reads sorted by length, streamed through LDS; + k_read_sums for the per-read chains, k_long_ea, k_long_gc_bins, k_adapter_first)", (bases, n, 2 * bases + 48 * n),
            lambda: FusedPass(QCMetrics(), AdapterCounter(list(synth.NANOPORE_PROBES))), c4_step,
            lambda f, passes: {"base_table_sum_ok": bool(int(np.array(f.qc_metrics.base_count_table(), np.uint64).sum()) == bases * passes),
                               "phred_scores_sum_ok": bool(int(np.array(f.qc_metrics.phred_scores(), np.uint64).sum()) == n * passes)})
        del arr
    # ---- the (a) modules the headline does not hold, alone and composed as the reference's driver loop composes them
    #      (__main__.py:279-306): every one over the HBM-resident 100 M reads of config 2 (pairs: config 3's), algorithmic
    #      bytes by SURVEY 8(d)'s per-module figures.  The estimator and the k-mer table are SETTLED when the timed passes
    #      start (one warm-up pass over all records went through them: the fresh object's pass is `first_pass_ms`) ----
    module_entries = ("single_end_six_modules", "overrep_alone", "dedup_single_end")
    if any(wanted(e) for e in module_entries):
        from sequali_amd import DedupEstimator, NanoStats, OverrepresentedSequences
        n, per = 100_000_000, 25_000_000
        batches = [synth.device_array(synth.ILLUMINA, k * per, per) for k in range(n // per)]
        bases = sum(b._batch.total_bases for b in batches)
        name_len = 36
        ovr_bytes = (2 * 5 * 21) * (n // 8) + 40 * n          # <= 210 B per sampled read (1 in 8) + 40 B/read
        dedup_bytes = 56 * n                                   # 16 B fingerprint + 40 B meta per read

        def timed_first(make, step):
            """(objects after one pass, ms of that first pass): what a fresh table / estimator costs"""
            o = make()
            _lib.synchronize()
            t0 = time.perf_counter()
            step(o)
            _lib.synchronize()
            return o, (time.perf_counter() - t0) * 1e3

        if wanted("overrep_alone"):
            def ovr_step(o):
                for b in batches:
                    o.add_record_array(b)
            first = {}

            def ovr_make():
                o, first["ms"] = timed_first(OverrepresentedSequences, ovr_step)
                return o
            out["overrep_alone"] = run(
                "overrep", f"{n} x 150 bp synthetic single-end reads, OverrepresentedSequences (defaults: every 8th read, 21-mers, 5 M uniques) "
                "alone, records resident in HBM; the table holds its 5 M fragments when the timed passes start",
                "k_overrep", (bases, n, ovr_bytes), ovr_make, ovr_step,
                lambda o, passes: {"number_of_sequences_ok": bool(o.number_of_sequences == n * (passes + 1)),
                                   "sampled_ok": bool(o.sampled_sequences == (n // 8) * (passes + 1)),
                                   "table_full_ok": bool(o.collected_unique_fragments == o.max_unique_fragments)})
            out["overrep_alone"]["first_pass_ms"] = round(first["ms"], 3)
        if wanted("dedup_single_end"):
            def dd_step(o):
                for b in batches:
                    o.add_record_array(b)
            first = {}

            def dd_make():
                o, first["ms"] = timed_first(lambda: DedupEstimator(front_sequence_offset=64, back_sequence_offset=0), dd_step)
                first["bits"] = o._modulo_bits
                return o
            out["dedup_single_end"] = run(
                "dedup", f"{n} x 150 bp synthetic single-end reads, DedupEstimator (the CLI's single-end geometry: 8 + 8 bases, front offset 64, "
                "back offset 0; 1 M fingerprints) alone, records resident in HBM; a settled estimator (one pass over all records in front)",
                "k_dedup_hash (+ table lookups on the device, the insertion tail of the new fingerprints on the host)",
                (bases, n, dedup_bytes), dd_make, dd_step,
                lambda o, passes: {"tracked_ok": bool(0 < o.tracked_sequences <= 1_000_000),
                                   "counts_sum_ok": bool(int(np.array(o.duplication_counts(), np.uint64).sum()) > 0)})
            out["dedup_single_end"]["first_pass_ms"] = round(first["ms"], 3)
            out["dedup_single_end"]["modulo_bits_after_first_pass"] = first["bits"]
        if wanted("single_end_six_modules"):
            def six_make():
                return (FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality()),
                        OverrepresentedSequences(), NanoStats(),
                        DedupEstimator(front_sequence_offset=64, back_sequence_offset=0))

            def six_step(o):
                fused, ovr, nano, dd = o
                for b in batches:      # __main__.py:279-306: metrics, adapters, per tile quality, overrepresented, nanostats, dedup
                    fused.add_record_array(b); clear(fused)
                    ovr.add_record_array(b)
                    nano.add_record_array(b)
                    dd.add_record_array(b)
            first = {}

            def six_make_settled():
                o, first["ms"] = timed_first(six_make, six_step)
                return o
            out["single_end_six_modules"] = run(
                "six", f"{n} x 150 bp synthetic single-end reads through the reference's driver loop (__main__.py:279-306): "
                "FusedPass(QCMetrics, AdapterCounter, PerTileQuality) + OverrepresentedSequences + NanoStats + DedupEstimator per array of 25 M, "
                "records resident in HBM (NanoStats skips at the first header that is not a nanopore one, as the reference does)",
                "the kernels `route` names",
                (bases, n, 2 * bases + (48 + name_len) * n + ovr_bytes + dedup_bytes), six_make_settled, six_step,
                lambda o, passes: {
                    "base_table_sum_ok": bool(int(np.array(o[0].qc_metrics.base_count_table(), np.uint64).sum()) == bases * (passes + 1)),
                    "pertile_reads_ok": bool(o[0].per_tile_quality.number_of_reads == n * (passes + 1)),
                    "overrep_sequences_ok": bool(o[1].number_of_sequences == n * (passes + 1)),
                    "nanostats_skipped_ok": bool(o[2].skipped_reason is not None),
                    "dedup_tracked_ok": bool(0 < o[3].tracked_sequences <= 1_000_000)})
            out["single_end_six_modules"]["first_pass_ms"] = round(first["ms"], 3)
        del batches
    if wanted("dedup_paired") or wanted("insert_size_alone"):
        from sequali_amd import DedupEstimator
        n, per = 100_000_000, 25_000_000
        r1 = [synth.device_array(synth.ILLUMINA, k * per, per) for k in range(n // per)]
        r2 = [synth.device_array(synth.ILLUMINA_R2, k * per, per) for k in range(n // per)]
        bases = sum(b._batch.total_bases for b in r1) + sum(b._batch.total_bases for b in r2)
        if wanted("dedup_paired"):
            def ddp_step(o):
                for a, b in zip(r1, r2):
                    o.add_record_array_pair(a, b)

            def ddp_make():
                o = DedupEstimator(front_sequence_offset=0, back_sequence_offset=0)
                ddp_step(o)
                return o
            out["dedup_paired"] = run(
                "dedup_paired", f"{n} x 150 bp synthetic pairs, DedupEstimator (the CLI's paired geometry: 8 bases of each mate at offset 0) alone, "
                "records resident in HBM; a settled estimator",
                "k_dedup_hash (+ table lookups on the device, the insertion tail of the new fingerprints on the host)",
                (bases, n, (16 + 80) * n), ddp_make, ddp_step,
                lambda o, passes: {"tracked_ok": bool(0 < o.tracked_sequences <= 1_000_000)})
        if wanted("insert_size_alone"):
            def isz_step(o):
                for a, b in zip(r1, r2):
                    o.add_record_array_pair(a, b)
            out["insert_size_alone"] = run(
                "insert_size", f"{n} x 150 bp synthetic pairs, InsertSizeMetrics alone, records resident in HBM",
                "the kernels `route` names",
                (bases, n, bases // 2 + (32 + 80) * n), InsertSizeMetrics, isz_step,   # 1 B/base of read 1 + 32 B of read 2's ends + 80
                lambda o, passes: {"insert_size_pairs_ok": bool(o.total_reads == n * passes),
                                   "insert_sizes_sum_ok": bool(int(np.array(o.insert_sizes(), np.uint64).sum()) == n * passes)})
        del r1, r2
    # ---- end to end from host memory (not HBM resident: host / PCIe bound, never `value`) ----
    if wanted("e2e_host_fastq_default_buffer") or wanted("e2e_pinned_64MiB_device_split") or wanted("e2e_host_fastq_default_buffer_six_modules"):
        import io
        from sequali_amd import FastqParser, PinnedReader
        n = 2_000_000
        text = synth.illumina_fastq(0, n)

        def e2e(make_file, six=False, **parser_kw):
            from sequali_amd import DedupEstimator, NanoStats, OverrepresentedSequences
            f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)), PerTileQuality() if six else None)
            more = (OverrepresentedSequences(), NanoStats(), DedupEstimator(front_sequence_offset=64, back_sequence_offset=0)) if six else ()
            fobj = make_file()
            _lib.synchronize()
            t0 = time.perf_counter()
            arrays = 0
            for a in FastqParser(fobj, **parser_kw):
                f.add_record_array(a)      # __main__.py:279-306: one call per module and array
                for mod in more:
                    mod.add_record_array(a)
                arrays += 1
            f.qc_metrics.flush()
            ok = bool(int(np.array(f.qc_metrics.base_count_table(), np.uint64).sum()) == 150 * n)
            if six:
                ok = ok and f.per_tile_quality.number_of_reads == n and more[0].number_of_sequences == n and more[2].tracked_sequences > 0
            _lib.synchronize()
            dt = time.perf_counter() - t0
            return dt, arrays, ok

        first_dt, _, _ = e2e(lambda: io.BytesIO(text))       # first pass: page-locks its 64 MiB staging blocks (they go to a pool)
        runs = [e2e(lambda: io.BytesIO(text)) for _ in range(3)]   # the reference's call pattern: default initial_buffersize
        dt, arrays, ok = sorted(runs)[1]                     # host bound, ~0.1 s a pass: the median of three
        ok = all(r[2] for r in runs)
        out["e2e_host_fastq_default_buffer"] = {
            "workload": f"{n} x 150 bp FASTQ text in host memory (io.BytesIO) through FastqParser at its default 128 KiB ({arrays} arrays, "
                        "~380 reads each), QCMetrics + AdapterCounter called once per array as __main__.py:279-306 does; the parser's buffer logic "
                        "runs in the C ABI over page-locked 64 MiB blocks (sq_feeder), whose worker threads read the BytesIO's buffer and note the "
                        "line ends, one upload and one launch per block; file read, record split, upload and counting included",
            "value": round(150 * n / dt / 1e9, 3), "unit": "Gbases/s", "seconds": round(dt, 3),
            "first_pass_seconds": round(first_dt, 3), "seconds_of_three_passes": [round(r[0], 3) for r in runs],
            "checks": {"base_table_sum_ok": ok}}
        if wanted("e2e_host_fastq_default_buffer_six_modules"):
            e2e(lambda: io.BytesIO(text), six=True)
            runs = [e2e(lambda: io.BytesIO(text), six=True) for _ in range(3)]
            dt, arrays, ok = sorted(runs)[1]
            out["e2e_host_fastq_default_buffer_six_modules"] = {
                "workload": f"the same {n} reads and call pattern ({arrays} arrays at the default 128 KiB) with all six single-end modules of __main__.py:279-306 "
                            "called once per array: FusedPass(QCMetrics, AdapterCounter, PerTileQuality), OverrepresentedSequences, NanoStats, DedupEstimator",
                "value": round(150 * n / dt / 1e9, 3), "unit": "Gbases/s", "seconds": round(dt, 3),
                "seconds_of_three_passes": [round(r[0], 3) for r in runs], "checks": {"tables_ok": all(r[2] for r in runs)}}
        big = dict(initial_buffersize=64 << 20, split_on_device=True)
        reader = PinnedReader(text)                            # the text in page-locked memory, as a file object

        def rewound():
            reader.seek(0)
            return reader

        e2e(rewound, **big)
        runs = [e2e(rewound, **big) for _ in range(3)]
        dt, arrays, ok = sorted(runs)[1]
        ok = all(r[2] for r in runs)
        out["e2e_pinned_64MiB_device_split"] = {
            "workload": f"the same {n} reads as text in page-locked host memory (PinnedReader) through FastqParser(initial_buffersize=64 MiB, "
                        f"split_on_device=True): {arrays} arrays, uploaded from where they lie, records split on the GPU (k_split_*), "
                        "QCMetrics + AdapterCounter; upload, split and counting included",
            "value": round(150 * n / dt / 1e9, 3), "unit": "Gbases/s", "seconds": round(dt, 3),
            "seconds_of_three_passes": [round(r[0], 3) for r in runs],
            "checks": {"base_table_sum_ok": ok}}
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("SQ_DEVICE", str(local_rank))   # tests put two ranks on one GPU
    device_index = int(os.environ["SQ_DEVICE"])

    dist = None
    torch = None
    use_dist = "WORLD_SIZE" in os.environ   # launched by torch.distributed.run, even with 1 rank
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        backend = os.environ.get("SQ_BENCH_BACKEND", "nccl")   # gloo: tests with both ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import numpy as np
    from sequali_amd import AdapterCounter, PerTileQuality, QCMetrics, _lib, synth
    lib, ctx = _lib.lib(), _lib.context()

    mods = set(args.modules.split(","))
    kind = synth.NANOPORE if args.kind == "nanopore" else synth.ILLUMINA
    if args.kind == "illumina" and os.environ.get("SQ_BENCH_BY_TILE"):  # the order a sequencer writes
        kind = synth.ILLUMINA_BY_TILE
    probes = synth.NANOPORE_PROBES if args.kind == "nanopore" else synth.ILLUMINA_PROBES
    qc = QCMetrics() if "qc" in mods else None
    ad = AdapterCounter(list(probes)) if "adapter" in mods else None
    pt = PerTileQuality() if "pertile" in mods else None

    # ---- the rank's shard, generated straight into HBM (not timed) ----
    first = rank * args.reads
    batches = []
    done = 0
    while done < args.reads:
        n = min(args.batch_reads, args.reads - done)
        batches.append(synth.device_array(kind, first + done, n))
        done += n
    total_bases = sum(b._batch.total_bases for b in batches)

    device = f"cuda:{device_index}"
    scratch = []
    reduce_ms = []
    reduce_events = []
    lib_stream = [torch.cuda.ExternalStream(int(lib.sq_stream_handle(ctx)), device=device)] if use_dist else [None]

    def step(events=None):
        for b in batches:
            if events is not None:
                events.start()
            _lib.check(lib.sq_fused_add_batch(b._batch.handle, qc._h if qc else None,
                                              ad._h if ad else None, pt._h if pt else None))
            if events is not None:
                events.stop()
        if not use_dist:
            _lib.synchronize()
        if use_dist:
            # the job's one exchange step: sum the count tables of all ranks over RCCL, as ONE
            # all-reduce over one flat buffer (on a copy, so that repeated steps keep accumulating
            # the local counts).  Set up once, behind the first pass: the ranks agree on the
            # job's longest read and longest adapter row first (their shards may differ), so
            # that every rank presents the same shapes
            from sequali_amd import dist as sqdist
            if not scratch:
                ml = sqdist.agree_on_shapes(qc, ad, device)
                tables = (sqdist.qcmetrics_tables(qc, device, ml) if qc else []) + \
                         (sqdist.adaptercounter_tables(ad, device) if ad else [])
                tables = [t.reshape(-1) for t in tables if t.dtype == torch.int64]
                flat = torch.empty(sum(t.numel() for t in tables), dtype=torch.int64, device=device)
                scratch.extend([tables, flat, list(flat.split([t.numel() for t in tables])), None])
            tables, flat, parts, _ = scratch
            # The pass ran on the library's stream, the copy and the collective run on torch's: each waits for the other ON THE
            # DEVICE (hipStreamWaitEvent both ways) -- the copy for the pass's kernels, the next pass for the copy -- so a
            # step costs no host round trip; the barrier behind the timed steps waits for everything.  (Until round 5 three
            # host-side synchronisations per step: ~ 1 ms of a 12 ms step.)
            cur = torch.cuda.current_stream()
            cur.wait_stream(lib_stream[0])
            torch._foreach_copy_(parts, tables)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            scratch[3] = sqdist._all_reduce(flat)   # in place over RCCL (through the host under gloo)
            ev1.record()
            lib_stream[0].wait_stream(cur)
            if events is not None:
                reduce_events.append((ev0, ev1))   # the collective alone (it runs on torch's stream); read behind the barrier

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        _lib.synchronize()

    lib.sq_route_reset(ctx)
    for _ in range(args.warmup):
        step()
    headline_route = compact_route((lib.sq_last_route(ctx) or b"").decode()) if args.warmup else None
    events = HipEvents(lib.sq_stream_handle(ctx))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = events.durations_ms()
    reduce_ms = [e0.elapsed_time(e1) for e0, e1 in reduce_events]

    job_bases = world * total_bases
    rank_ms = None
    if use_dist:
        from sequali_amd import dist as sqdist
        # where the time goes, rank by rank: ms per step, of which kernels, of which the collective
        mine = torch.tensor([elapsed / args.steps * 1e3, sum(launch_ms) / args.steps, sum(reduce_ms) / max(len(reduce_ms), 1)],
                            dtype=torch.float64, device=device)
        mine = sqdist._wire(mine)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [[round(float(x), 3) for x in t.tolist()] for t in every]
        elapsed = float(sqdist._all_reduce(torch.tensor([elapsed], dtype=torch.float64, device=device), dist.ReduceOp.MAX).item())
        job_bases = int(sqdist._all_reduce(torch.tensor([total_bases], dtype=torch.int64, device=device)).item())

    # ---- size-independent checks at full size ----
    passes = args.warmup + args.steps
    checks = {}
    if qc is not None:
        base = np.array(qc.base_count_table(), dtype=np.uint64)
        phred = np.array(qc.phred_count_table(), dtype=np.uint64)
        checks["base_table_sum_ok"] = bool(int(base.sum()) == total_bases * passes)
        checks["phred_table_sum_ok"] = bool(int(phred.sum()) == total_bases * passes)
        if args.kind == "illumina":
            checks["per_position_ok"] = bool((base.reshape(-1, 5).sum(axis=1) == args.reads * passes).all())
        checks["gc_hist_sum_ok"] = bool(int(np.array(qc.gc_content(), dtype=np.uint64).sum())
                                        <= args.reads * passes)
        checks["phred_scores_sum_ok"] = bool(int(np.array(qc.phred_scores(), dtype=np.uint64).sum())
                                             == args.reads * passes)
        if use_dist and scratch and passes > 0:
            # what the last step's all-reduce delivered: the base table of the whole job
            reduced = scratch[3][:scratch[0][0].numel()]   # the first table of the flat buffer is base_counts
            checks["reduced_base_table_sum_ok"] = bool(int(reduced.sum().item()) == job_bases * passes)
    if ad is not None:
        counts = ad.get_counts()
        checks["adapter_fwd_eq_rev_ok"] = bool(all(
            int(np.array(f, dtype=np.uint64).sum()) == int(np.array(r, dtype=np.uint64).sum())
            for _, f, r in counts))

    if rank == 0:
        value = job_bases * args.steps / elapsed / 1e9
        avg_ms = sum(launch_ms) / max(len(launch_ms), 1)
        reads_per_launch = sum(len(b) for b in batches) / len(batches)
        # 2 B/base + 40 B/read meta + 8 B/read error-rate write-back
        algo_bytes = (2 * total_bases + 48 * args.reads) / len(batches)
        traffic = None
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):   # PMC passes cannot run inside this process; see profiles/README.md
            with open(tpath) as f:
                tj = json.load(f)
            # only a measurement of THIS build counts: the file names the kernel sources it was taken on
            if tj.get("reads_per_launch") == int(reads_per_launch) and tj.get("kind") == args.kind \
                    and tj.get("modules") == sorted(mods) and tj.get("csrc_sha") == csrc_sha():
                traffic = tj.get("hbm_bytes_per_launch")
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Gbases/s processed (1/2/4/8 GPU) + achieved HBM GB/s fraction",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{args.reads} x {READ_LEN if args.kind == 'illumina' else '~10 kb'} bp synthetic single-end {args.kind} FASTQ per GPU, "
                                   f"{'+'.join(sorted(mods))} fused pass, records resident in HBM",
                       "reads_per_gpu": args.reads, "reads_per_launch": int(reads_per_launch),
                       "modules": sorted(mods), "sharding": f"records x{world}, RCCL all-reduce of count tables"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": dominant_kernel(args.kind, mods),
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(launch_ms)},
            "route": headline_route,
            "checks": checks,
        }
        if rank_ms is not None:
            out["allreduce_ms"] = round(max(r[2] for r in rank_ms), 4)   # the all-reduce of the count tables alone, slowest rank
            out["per_rank_ms"] = {"columns": ["ms_per_step", "kernel_ms_per_step", "allreduce_ms"], "ranks": rank_ms}
        if world == 1 and not use_dist and args.other_configs and args.kind == "illumina":
            del batches[:]
            try:
                out["other_configs"] = other_configs(lib, ctx, max(1, min(args.steps, 3)), 1,
                                                     set(args.configs.split(",")) if args.configs else None)
                # HBM-side bytes per pass of every config (PMC passes of scripts/profile_r3.sh), under the same rule
                # as the headline's: only a measurement of THIS build of the kernels counts
                if os.path.exists(tpath) and tj.get("csrc_sha") == csrc_sha():
                    for name, b in (tj.get("other_configs_hbm_bytes_per_step") or {}).items():
                        if name in out["other_configs"] and "roofline" in out["other_configs"][name]:
                            out["other_configs"][name]["roofline"]["traffic"] = b
                try:
                    for name, base in cpu_baselines_other(set(out["other_configs"])).items():
                        if name in out["other_configs"]:
                            out["other_configs"][name]["cpu_baseline"] = base
                except Exception as e:   # a report, never a reason to lose the line
                    out["other_configs"]["cpu_baseline_error"] = repr(e)
            except Exception as e:   # never a reason to lose the headline
                out["other_configs"] = {"error": repr(e)}
        if world == 1 and args.cpu_sample > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, args.reads), max(1, args.cpu_passes))
            except Exception as e:  # the baseline is a report, never a reason to lose the line
                out["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
