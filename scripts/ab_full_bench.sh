cd "${GRAFT_REPO_ROOT:?}" || exit 1
for i in 1 2; do for n in base nn; do
SQ_LIB=$PWD/scripts/build/libsqgpu_$n.so timeout 600 python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d['other_configs']
print('$n', d['value'], d['roofline']['avg_launch_ms'], ' '.join(f\"{k}={o[k].get('ms_per_step')}\" for k in ('uniform_200bp','uniform_250bp','ragged_50_150','config3_paired','config3_paired_by_tile','config4_nanopore','overrep_alone','dedup_single_end','single_end_six_modules','insert_size_alone')))"
done; done
