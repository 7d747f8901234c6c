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
    ap.add_argument("--cpu-passes", type=int, default=6,
                    help="times the CPU baseline walks its sample (about 10 s of CPU work by default)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000,
                    help="records of the CPU baseline sample (0 = skip)")
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
        return "k_seg (long reads: k_read_sums + k_seg + k_adapter_first)"
    if "pertile" in mods:
        return "k_pass (fused per-base pass, tile-sorted order)"
    if "adapter" in mods and "qc" in mods:
        return "k_wide<AD> (fused per-base pass, one read length)"
    if "qc" in mods:
        return "k_ring (QCMetrics, one read length)"
    return "k_pass (fused per-base pass)"


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
    return {"value": round(bases * passes / dt / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": kind,
            "sample": f"first {sample_reads} records of the workload ({bases} bases) x {passes} passes, "
                      f"QCMetrics+AdapterCounter, {dt:.2f} s"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["SQ_DEVICE"] = str(local_rank)

    dist = None
    torch = None
    use_dist = "WORLD_SIZE" in os.environ   # launched by torch.distributed.run, even with 1 rank
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

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

    device = f"cuda:{local_rank}"
    scratch = []

    def step(events=None):
        for b in batches:
            if events is not None:
                events.start()
            _lib.check(lib.sq_fused_add_batch(b._batch.handle, qc._h if qc else None,
                                              ad._h if ad else None, pt._h if pt else None))
            if events is not None:
                events.stop()
        _lib.synchronize()
        if use_dist:
            # the job's one exchange step: sum the count tables of all ranks over RCCL, as ONE
            # all-reduce over one flat buffer (on a copy, so that repeated steps keep
            # accumulating the local counts); the aliases and the buffer are set up once
            if not scratch:
                from sequali_amd import dist as sqdist
                tables = (sqdist.qcmetrics_tables(qc, device) if qc else []) + \
                         (sqdist.adaptercounter_tables(ad, device) if ad else [])
                tables = [t.reshape(-1) for t in tables if t.dtype == torch.int64]
                flat = torch.empty(sum(t.numel() for t in tables), dtype=torch.int64, device=device)
                scratch.extend([tables, flat, list(flat.split([t.numel() for t in tables]))])
            tables, flat, parts = scratch
            torch._foreach_copy_(parts, tables)
            dist.all_reduce(flat)
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        _lib.synchronize()

    for _ in range(args.warmup):
        step()
    events = HipEvents(lib.sq_stream_handle(ctx))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = events.durations_ms()

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
    if ad is not None:
        counts = ad.get_counts()
        checks["adapter_fwd_eq_rev_ok"] = bool(all(
            int(np.array(f, dtype=np.uint64).sum()) == int(np.array(r, dtype=np.uint64).sum())
            for _, f, r in counts))

    if rank == 0:
        value = world * total_bases * args.steps / elapsed / 1e9
        avg_ms = sum(launch_ms) / max(len(launch_ms), 1)
        reads_per_launch = sum(len(b) for b in batches) / len(batches)
        # 2 B/base + 40 B/read meta + 8 B/read error-rate write-back
        algo_bytes = (2 * total_bases + 48 * args.reads) / len(batches)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):   # PMC passes cannot run inside this process; see profiles/README.md
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("reads_per_launch") == int(reads_per_launch) and tj.get("kind") == args.kind \
                    and tj.get("modules") == sorted(mods):
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
            "checks": checks,
        }
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
