for r in 250000 500000 1000000 2000000 10000000; do
  timeout 100 python bench.py --reads $r --batch-reads $r --cpu-sample 0 --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['reads_per_gpu'], d['value'], d['roofline']['avg_launch_ms'])"
done
