#!/bin/bash
# the PerTileQuality ride with reads of random tiles at the bench's size (bounded: 4 minutes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pair
timeout 240 python - <<'PY' 2>&1 | tail -20
import time
from sequali_amd import FusedPass, PerTileQuality, QCMetrics, synth, _lib
from sequali_amd._lib import context, lib
n = 25_000_000
ds = [synth.device_array(synth.ILLUMINA, 0, n), synth.device_array(synth.ILLUMINA_R2, 0, n)]
fs = [FusedPass(QCMetrics(), None, PerTileQuality()) for _ in ds]
for i in range(4):
    for k in range(2):
        lib().sq_route_reset(context())
        _lib.synchronize(); t0 = time.perf_counter()
        fs[k].add_record_array(ds[k]); fs[k].qc_metrics._pending.clear()
        _lib.synchronize(); dt = time.perf_counter() - t0
        print(i, k, f"{dt*1e3:.2f} ms", fs[k].per_tile_quality.number_of_reads, fs[k].per_tile_quality.skipped_reason, (lib().sq_last_route(context()) or b"").decode(), flush=True)
PY
