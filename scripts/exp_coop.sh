for m in qc qc,adapter; do for nc in 0 1; do
if [ $nc = 1 ]; then export SQ_NO_COOP=1; else unset SQ_NO_COOP; fi
timeout 200 python bench.py --modules $m --reads 50000000 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('modules=$m no_coop=$nc', d['value'], d['roofline']['avg_launch_ms'])"
done; done
