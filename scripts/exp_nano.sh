for br in 100000 400000 1000000; do
timeout 200 python bench.py --kind nanopore --reads 1000000 --batch-reads $br --cpu-sample 0 --steps 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch=$br', d['value'], d['roofline']['avg_launch_ms'], d['checks'])"
done
