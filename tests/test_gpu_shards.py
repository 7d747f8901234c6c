"""a job split over shards gives what one sequential run gives.

Every test feeds the same records (a) to the oracle in one piece and (b) to three shard
objects of the HIP path (uneven contiguous ranges, two batches each), merges them with
sequali_amd.dist and compares every getter.  In-process the merge runs without a process
group; test_two_processes_* runs it between two ranks (gloo, both on cuda:0)."""
import os
import socket
import warnings

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CUTS = (0.0, 0.23, 0.71, 1.0)


def u64(a):
    return np.array(a, dtype=np.uint64)


def spans(n, cuts=CUTS):
    edges = [int(round(c * n)) for c in cuts]
    return list(zip(edges[:-1], edges[1:]))


def batches(first, last):
    """two uneven batches of [first, last)"""
    mid = first + (last - first) // 3
    return [(a, b) for a, b in ((first, mid), (mid, last)) if b > a]


def sub_array(buf, metas, a, b):
    from sequali_amd import FastqRecordArrayView
    return FastqRecordArrayView._from_buffer(buf, metas[a:b].copy())


def random_reads(seed, n, max_len, alphabet=b"ACGT", tiles=(1101, 1102, 2205, 7), bad_at=None):
    rng = np.random.default_rng(seed)
    names, seqs, quals = [], [], []
    alpha = np.frombuffer(alphabet, np.uint8)
    pool = []
    for i in range(n):
        L = int(rng.integers(0, max_len + 1))
        if pool and rng.random() < 0.3:
            s = pool[int(rng.integers(0, len(pool)))]
            L = len(s)
        else:
            s = rng.choice(alpha, size=L).tobytes().decode()
            if len(pool) < 200:
                pool.append(s)
        q = (rng.integers(0, 42, size=L) + 33).astype(np.uint8).tobytes().decode()
        tile = int(rng.choice(tiles))
        name = f"M:1:F:{i % 4}:{tile}:{i}:{L} 1:N:0:X"
        if bad_at is not None and i in bad_at:
            name = f"read{i} without tile"
        names.append(name)
        seqs.append(s)
        quals.append(q)
    return oracle.make_batch(names, seqs, quals)


@pytest.mark.parametrize("kw", [dict(max_unique_fragments=900, sample_every=3),
                                dict(max_unique_fragments=150, sample_every=1, fragment_length=11),
                                dict(sample_every=7)])
def test_overrepresented_shards_equal_one_run(kw):
    from sequali_amd import OverrepresentedSequences, dist
    buf, metas = random_reads(61, 9000, 150, alphabet=b"ACGTACGTACGTN")
    ref = oracle.OverrepresentedSequences(**kw)
    ref.add(buf, metas)
    shards = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for first, last in spans(len(metas)):
            o = OverrepresentedSequences(**kw)
            o.set_shard(first)
            for a, b in batches(first, last):
                o.add_record_array(sub_array(buf, metas, a, b))
            shards.append(o)
    dist.merge_overrepresented(shards, DEV)
    for o in shards:
        assert o.number_of_sequences == ref.number_of_sequences == len(metas)
        assert o.sampled_sequences == ref.sampled_sequences
        assert o.total_fragments == ref.total_fragments
        assert o.collected_unique_fragments == ref.collected_unique_fragments
        assert o.sequence_counts() == ref.sequence_counts()
        assert o.overrepresented_sequences(0.001) == ref.overrepresented_sequences(0.001)
    if "max_unique_fragments" in kw:
        assert ref.collected_unique_fragments == kw["max_unique_fragments"]


def test_overrepresented_shard_table_grows():
    """more distinct fragments in one shard than the default table holds: the uncapped shard
    table doubles; the merged result still is the capped job table"""
    from sequali_amd import OverrepresentedSequences, dist, synth
    kw = dict(max_unique_fragments=2000, sample_every=1)
    n = 30000
    ref = oracle.OverrepresentedSequences(**kw)
    shards = []
    for first, last in spans(n, (0.0, 0.5, 1.0)):
        arr = synth.host_array(synth.ILLUMINA, first, last - first)
        ref.add(arr.obj, arr._metas)
        o = OverrepresentedSequences(**kw)
        o.set_shard(first)
        o.add_record_array(arr)
        assert o.collected_unique_fragments > 50000   # shard tables are not capped
        shards.append(o)
    dist.merge_overrepresented(shards, DEV)
    assert shards[1].collected_unique_fragments == 2000
    assert shards[1].sequence_counts() == ref.sequence_counts()
    assert shards[0].total_fragments == ref.total_fragments


@pytest.mark.parametrize("method", ["relay", "gather"])
@pytest.mark.parametrize("paired,max_len", [(False, 90), (True, 90), (True, 14)])
def test_dedup_shards_equal_one_run(paired, max_len, method):
    """max_len 14: pairs shorter than the fingerprint, whose hashes show bytes of the pair
    in front -- also across a shard boundary (:4512-4516).  method "gather": every shard counts
    its lower bound (sort + histogram on the device) and filters its hashes, the first shard
    runs the one insertion tail (tests/test_dedup_gather_cpu.py has the host halves)"""
    from sequali_amd import DedupEstimator, dist
    kw = dict(max_stored_fingerprints=250, front_sequence_offset=4, back_sequence_offset=0)
    n = 12000
    b1, m1 = random_reads(71, n, max_len)
    b2, m2 = random_reads(72, n, max_len)
    ref = oracle.DedupEstimator(**kw)
    if paired:
        ref.add_pair(b1, m1, b2, m2)
    else:
        ref.add(b1, m1)
    shards = []
    for first, last in spans(n):
        d = DedupEstimator(**kw)
        d.set_deferred(True)
        for a, b in batches(first, last):
            if paired:
                d.add_record_array_pair(sub_array(b1, m1, a, b), sub_array(b2, m2, a, b))
            else:
                d.add_record_array(sub_array(b1, m1, a, b))
        shards.append(d)
    dist.merge_dedup(shards, DEV, method=method)
    assert ref._modulo_bits >= 3
    for d in shards:
        assert d._modulo_bits == ref._modulo_bits
        assert d.tracked_sequences == ref.tracked_sequences
        np.testing.assert_array_equal(u64(d.duplication_counts()), ref.duplication_counts())


@pytest.mark.parametrize("max_adapters", [12, 10000])
def test_insertsize_shards_equal_one_run(max_adapters):
    from sequali_amd import InsertSizeMetrics, dist, synth
    n = 24000
    ref = oracle.InsertSizeMetrics(max_adapters)
    shards = []
    for first, last in spans(n):
        z = InsertSizeMetrics(max_adapters)
        z.set_shard(first, 19)
        for a, b in batches(first, last):
            a1 = synth.host_array(synth.ILLUMINA, a, b - a)
            a2 = synth.host_array(synth.ILLUMINA_R2, a, b - a)
            ref.add_pair(a1.obj, a1._metas, a2.obj, a2._metas)
            z.add_record_array_pair(a1, a2)
        shards.append(z)
    dist.merge_insertsize(shards, DEV)
    for z in shards:
        assert z.total_reads == ref.total_reads == n
        assert z.number_of_adapters_read1 == ref.number_of_adapters_read1 > 0
        assert z.number_of_adapters_read2 == ref.number_of_adapters_read2 > 0
        np.testing.assert_array_equal(u64(z.insert_sizes()), ref.insert_sizes())
        assert z.adapters_read1() == ref.adapters_read1()    # slot order included
        assert z.adapters_read2() == ref.adapters_read2()


@pytest.mark.parametrize("bad_at", [None, {5200}, {100, 8000}])
def test_pertile_shards_equal_one_run(bad_at):
    from sequali_amd import PerTileQuality, dist
    n = 9000
    buf, metas = random_reads(81, n, 120, tiles=(1101, 1102, 2205, 7, 99239, 31), bad_at=bad_at)
    ref = oracle.PerTileQuality()
    ref.add(buf, metas)
    shards, firsts = [], []
    for first, last in spans(n):
        p = PerTileQuality()
        for a, b in batches(first, last):
            p.add_record_array(sub_array(buf, metas, a, b))
        shards.append(p)
        firsts.append(first)
    dist.merge_pertile(shards, firsts, DEV)
    for p in shards:
        assert p.number_of_reads == ref.number_of_reads == (n if bad_at is None else min(bad_at))
        assert p.max_length == ref.max_length
        if bad_at is None:
            assert p.skipped_reason is None and not ref.skipped
        else:
            assert ref.skipped and ref.skipped_record == min(bad_at)
            assert p.skipped_reason == f"Can not parse header: 'read{min(bad_at)} without tile'"
        got, want = p.get_tile_counts(), ref.get_tile_counts()
        assert [t for t, _, _ in got] == [t for t, _, _ in want]
        for (t, e, c), (tr, er, cr) in zip(got, want):
            np.testing.assert_allclose(np.array(e), er, rtol=1e-6)
            np.testing.assert_array_equal(u64(c), cr)


# ---- two processes -------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, n, out_dir, dedup_method="relay"):
    import torch.distributed as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SQ_DEVICE="0")
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from sequali_amd import (DedupEstimator, InsertSizeMetrics, OverrepresentedSequences, PerTileQuality, dist,
                             synth)
    first, last = dist.shard_range(n, rank, world)
    a1 = synth.host_array(synth.ILLUMINA, first, last - first)
    a2 = synth.host_array(synth.ILLUMINA_R2, first, last - first)
    o = OverrepresentedSequences(max_unique_fragments=3000, sample_every=2)
    o.set_shard(first)
    o.add_record_array(a1)
    d = DedupEstimator(max_stored_fingerprints=400)
    d.set_deferred(True)
    d.add_record_array_pair(a1, a2)
    z = InsertSizeMetrics(40)
    z.set_shard(first, 19)
    z.add_record_array_pair(a1, a2)
    p = PerTileQuality()
    p.add_record_array(a1)
    dist.merge_overrepresented([o], DEV)
    dist.merge_dedup([d], DEV, method=dedup_method)
    dist.merge_insertsize([z], DEV)
    dist.merge_pertile([p], [first], DEV)
    tiles = p.get_tile_counts()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             ov_keys=np.array(sorted(o.sequence_counts().items()), dtype=object),
             ov_scalars=u64([o.number_of_sequences, o.sampled_sequences, o.total_fragments,
                             o.collected_unique_fragments]),
             dd=u64(d.duplication_counts()), dd_scalars=u64([d._modulo_bits, d.tracked_sequences]),
             isz=u64(z.insert_sizes()), isz1=np.array(z.adapters_read1(), dtype=object),
             isz2=np.array(z.adapters_read2(), dtype=object),
             isz_scalars=u64([z.total_reads, z.number_of_adapters_read1, z.number_of_adapters_read2]),
             pt_tiles=np.array([t for t, _, _ in tiles]), pt_err=np.array([e for _, e, _ in tiles]),
             pt_cnt=u64([c for _, _, c in tiles]), pt_reads=p.number_of_reads)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world,dedup_method", [(2, "relay"), (4, "relay"), (4, "gather"), (3, "gather")])
def test_processes_merge_equals_one_run(tmp_path, world, dedup_method):
    """world 4: the DedupEstimator relay (or gather) and the candidate selection of OverrepresentedSequences /
    InsertSizeMetrics run over more than one hop (four processes on cuda:0, gloo)"""
    import torch.multiprocessing as mp
    from sequali_amd import synth
    n = 20000
    mp.spawn(_rank_main, args=(world, _free_port(), n, str(tmp_path), dedup_method), nprocs=world, join=True)
    b1, m1 = synth.host_records(synth.ILLUMINA, 0, n)
    b2, m2 = synth.host_records(synth.ILLUMINA_R2, 0, n)
    o = oracle.OverrepresentedSequences(max_unique_fragments=3000, sample_every=2)
    o.add(b1, m1)
    d = oracle.DedupEstimator(max_stored_fingerprints=400)
    d.add_pair(b1, m1, b2, m2)
    z = oracle.InsertSizeMetrics(40)
    z.add_pair(b1, m1, b2, m2)
    p = oracle.PerTileQuality()
    p.add(b1, m1)
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz", allow_pickle=True)
        assert dict((k, int(v)) for k, v in g["ov_keys"]) == o.sequence_counts()
        assert g["ov_scalars"].tolist() == [o.number_of_sequences, o.sampled_sequences, o.total_fragments,
                                            o.collected_unique_fragments]
        np.testing.assert_array_equal(g["dd"], d.duplication_counts())
        assert g["dd_scalars"].tolist() == [d._modulo_bits, d.tracked_sequences]
        np.testing.assert_array_equal(g["isz"], z.insert_sizes())
        assert [(a, int(c)) for a, c in g["isz1"]] == z.adapters_read1()
        assert [(a, int(c)) for a, c in g["isz2"]] == z.adapters_read2()
        assert g["isz_scalars"].tolist() == [z.total_reads, z.number_of_adapters_read1, z.number_of_adapters_read2]
        want = p.get_tile_counts()
        assert g["pt_tiles"].tolist() == [t for t, _, _ in want]
        np.testing.assert_allclose(g["pt_err"], np.array([e for _, e, _ in want]), rtol=1e-6)
        np.testing.assert_array_equal(g["pt_cnt"], u64([c for _, _, c in want]))
        assert int(g["pt_reads"]) == p.number_of_reads


# ---- the config-5 collective on the device tables, two ranks -------------------------------
def _rank_tables(rank, world, port, n, out_dir):
    """config-2 modules on this rank's shard of 200 k synthetic reads cut to ragged lengths (the
    ranks then hold different max_length and different AdapterCounter row lengths: rank 1 sees
    a short batch first, so its rows were grown geometrically), then the merges on the DEVICE
    tables"""
    import torch.distributed as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SQ_DEVICE="0")
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from sequali_amd import AdapterCounter, FusedPass, QCMetrics, _lib, dist, synth
    first, last = dist.shard_range(n, rank, world)
    qc, ad = QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES))
    f = FusedPass(qc, ad)
    cut = first + (last - first) // 3
    for k, (a, b) in enumerate(((first, cut), (cut, last))):
        arr = synth.device_array(synth.ILLUMINA, a, b - a)
        # rank 0: 150 then at most 120 bases; rank 1: at most 100 then at most 140
        lo, trim = [(150, None), (60, 120)][k] if rank == 0 else [(40, 100), (70, 140)][k]
        if trim is not None:
            _lib.check(_lib.lib().sq_synth_trim(arr._batch.handle, 1000 + a, lo))
            # cap at `trim`: a second cut of the reads that are still longer
            buf, metas = arr._batch.download()
            metas = metas.copy()
            metas["sequence_length"] = np.minimum(metas["sequence_length"], trim)
            from sequali_amd import FastqRecordArrayView
            arr = FastqRecordArrayView._from_buffer(buf, metas)
            lengths = metas["sequence_length"]
        else:
            lengths = arr._batch.download()[1]["sequence_length"]
        np.save(os.path.join(out_dir, f"len_{a}.npy"), lengths)
        f.add_record_array(arr)
    qc.flush()
    rows_before = int(_lib.lib().sq_adaptercounter_row_length(ad._h))
    dist.merge_qcmetrics(qc, DEV)
    dist.merge_adaptercounter(ad, DEV)
    np.savez(os.path.join(out_dir, f"tables{rank}.npz"), base=u64(qc.base_count_table()), phred=u64(qc.phred_count_table()),
             ea_base=u64(qc.end_anchored_base_count_table()), ea_phred=u64(qc.end_anchored_phred_count_table()),
             gc=u64(qc.gc_content()), ps=u64(qc.phred_scores()), reads=qc.number_of_reads, ml=qc.max_length,
             fwd=np.array([u64(fw) for _, fw, _ in ad.get_counts()]), rev=np.array([u64(r) for _, _, r in ad.get_counts()]),
             seqs=ad.number_of_sequences, rows_before=rows_before)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_processes_sum_the_device_tables(tmp_path, world):
    """dist.merge_qcmetrics + dist.merge_adaptercounter between two / four
    ranks on the tables in HBM (gloo, all ranks on cuda:0) equal the oracle's single run bit for
    bit, with unequal max_length and unequal AdapterCounter row lengths across the ranks"""
    import torch.multiprocessing as mp
    from sequali_amd import synth
    n = 200_000
    mp.spawn(_rank_tables, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    buf, metas = synth.host_records(synth.ILLUMINA, 0, n)
    metas = metas.copy()
    for f in sorted(os.listdir(tmp_path)):
        if f.startswith("len_"):
            a = int(f[4:-4])
            L = np.load(tmp_path / f)
            metas["sequence_length"][a:a + len(L)] = L
    rq, ra = oracle.QCMetrics(), oracle.AdapterCounter(list(synth.ILLUMINA_PROBES))
    rq.add(buf, metas)
    ra.add(buf, metas)
    rows = []
    for rank in range(world):
        g = np.load(tmp_path / f"tables{rank}.npz")
        rows.append(int(g["rows_before"]))
        assert int(g["reads"]) == rq.number_of_reads and int(g["ml"]) == rq.max_length == 150
        np.testing.assert_array_equal(g["base"], rq.base_count_table())
        np.testing.assert_array_equal(g["phred"], rq.phred_count_table())
        np.testing.assert_array_equal(g["ea_base"], rq.end_anchored_base_count_table())
        np.testing.assert_array_equal(g["ea_phred"], rq.end_anchored_phred_count_table())
        np.testing.assert_array_equal(g["gc"], rq.gc_content())
        np.testing.assert_array_equal(g["ps"], rq.phred_scores())
        assert int(g["seqs"]) == ra.number_of_sequences
        for i, (_, fw, rv) in enumerate(ra.get_counts()):
            np.testing.assert_array_equal(g["fwd"][i], fw)
            np.testing.assert_array_equal(g["rev"][i], rv)
    assert rows[0] != rows[1], "the ranks were meant to hold different AdapterCounter row lengths"


def test_bench_step_with_two_ranks(tmp_path):
    """bench.py's N > 1 step (table shapes agreed first, ONE all-reduce over a flat copy of the
    device tables) run by two ranks on one GPU through torch.distributed.run (gloo instead of
    RCCL: RCCL does not take two ranks on one device); the reduced base table must hold the bases
    of both shards, and the line must come out with n_gpus 2"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SQ_DEVICE="0", SQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "300000", "--batch-reads", "200000", "--cpu-sample", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert all(d["checks"].values()), d["checks"]
    assert "reduced_base_table_sum_ok" in d["checks"]
    # where the time goes on every rank, and the collective alone
    assert d["allreduce_ms"] > 0 and len(d["per_rank_ms"]["ranks"]) == 2
    assert all(r[0] >= r[1] > 0 for r in d["per_rank_ms"]["ranks"])
