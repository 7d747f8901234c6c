# how sensitive is the fused pass to waves per CU?  (extra LDS per workgroup lowers residency)
for pad in 0 16000 42000 60000; do
SQ_LDS_PAD=$pad timeout 100 python bench.py --reads 20000000 --batch-reads 10000000 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds_pad=$pad', d['value'], d['roofline']['avg_launch_ms'])"
done
