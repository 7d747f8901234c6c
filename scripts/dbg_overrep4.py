"""4 in-process shards of the multi-process test's job: where do the merged counts go wrong?"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from oracle import oracle
from sequali_amd import OverrepresentedSequences, dist, synth
DEV = torch.device("cuda:0")
n = 20000
kw = dict(max_unique_fragments=3000, sample_every=2)
for world in (2, 4):
    b1, m1 = synth.host_records(synth.ILLUMINA, 0, n)
    ref = oracle.OverrepresentedSequences(**kw)
    ref.add(b1, m1)
    shards, solo = [], []
    for rank in range(world):
        first, last = dist.shard_range(n, rank, world)
        arr = synth.host_array(synth.ILLUMINA, first, last - first)
        o = OverrepresentedSequences(**kw)
        o.set_shard(first)
        o.add_record_array(arr)
        shards.append(o)
        # the shard alone, uncapped, through the oracle
        r = oracle.OverrepresentedSequences(max_unique_fragments=10**7, sample_every=2)
        # sampling follows the job-wide index: first is even for these splits
        r.add(arr.obj, arr._metas)
        solo.append(r.sequence_counts())
        got = o.sequence_counts()
        bad = {k: (got.get(k), v) for k, v in solo[-1].items() if got.get(k) != v}
        print(f"world {world} rank {rank}: first {first}, shard table {len(got)} keys, oracle {len(solo[-1])}, differing {len(bad)}", list(bad.items())[:3])
    dist.merge_overrepresented(shards, DEV)
    want = ref.sequence_counts()
    got = shards[0].sequence_counts()
    bad = {k: (got.get(k), v) for k, v in want.items() if got.get(k) != v}
    print(f"world {world}: merged differing {len(bad)} of {len(want)}", list(bad.items())[:5])
    for k in list(bad)[:5]:
        print("   ", k, [s.get(k) for s in solo])
