for t in 0 1; do for br in 10000000 25000000; do
  SQ_TOUCH=$t timeout 200 python bench.py --reads 50000000 --batch-reads $br --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('touch=$t batch=$br', d['value'], d['roofline']['avg_launch_ms'])"
done; done
