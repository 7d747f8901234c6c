#!/bin/bash
cd $GRAFT_REPO_ROOT
for skip in "" "o" "d" "dp" "z" "o,d" "dp,z" "o,d,dp,z"; do
  echo "== skip [$skip]"; FUZZ_SKIP=$skip timeout 300 python -u scripts/fuzz.py 5 55 4 2>&1 | grep -v amdgpu | cut -c1-120
done
echo "== SQ_STAGE_BYTES=0"; SQ_STAGE_BYTES=0 timeout 300 python -u scripts/fuzz.py 5 55 4 2>&1 | grep -v amdgpu | cut -c1-120
