#!/bin/bash
# a long randomised differential run on one box: scripts/fuzz.py over several seeds, one line per seed
cd "${GRAFT_REPO_ROOT:?}" || exit 1
OUT=gpurun_out/fuzz
mkdir -p $OUT
: > $OUT/summary.txt
for seed in ${SEEDS:-21 22 23 24 25 26 27 28}; do
  timeout ${PER_SEED:-420} python scripts/fuzz.py ${ITERS:-150} $seed > $OUT/seed$seed.log 2>&1
  echo "seed $seed rc=$? $(grep -c '\] ok  ' $OUT/seed$seed.log) ok, $(grep -c 'FAIL' $OUT/seed$seed.log) FAIL, $(tail -1 $OUT/seed$seed.log | cut -c1-100)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
