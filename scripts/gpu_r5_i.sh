#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 900 python -u scripts/fuzz.py 200 55 > gpurun_out/r5i/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r5i/fuzz.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs > gpurun_out/r5i/bench_torchrun1.json 2> gpurun_out/r5i/bench_torchrun1.err; echo "torchrun rc=$?"; tail -c 1500 gpurun_out/r5i/bench_torchrun1.json; tail -3 gpurun_out/r5i/bench_torchrun1.err
