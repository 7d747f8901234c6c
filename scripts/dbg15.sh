cd $GRAFT_REPO_ROOT
for e in "SQ_PT_FUSED=0" "SQ_SPAN=0" "SQ_SPAN_SPLIT=0" "FUZZ_SKIP=z" "X=1"; do
  echo "== $e"; env $e timeout 300 python scripts/fuzz.py 30 4 15 2>&1 | tail -2
done
