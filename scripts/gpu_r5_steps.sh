#!/bin/bash
# usage: gpurun --timeout N -- 'bash scripts/gpu_r5_steps.sh 1 2 3 ...'   -- the given steps of scripts/gpu_bisect_r5.sh in ONE
# call, each under its own timeout, after writing down what the box is (host memory, cores, whether /proc gives the watchdog
# what it needs).  A step that ends with the watchdog's code 97 stops the list.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bisect
{ grep -E "MemTotal|MemAvailable|CommitLimit" /proc/meminfo; nproc; cat /proc/sys/vm/overcommit_memory;
  python -c "import psutil; print('psutil', psutil.__version__)" 2>&1 | tail -1; } > gpurun_out/bisect/box.txt 2>&1
cat gpurun_out/bisect/box.txt
for k in "$@"; do
  bash scripts/gpu_bisect_r5.sh $k
  if grep -q "ending the test run before the machine does" gpurun_out/bisect/step$k.log 2>/dev/null; then echo "WATCHDOG fired in step $k"; exit 97; fi
done
