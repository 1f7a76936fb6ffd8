R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04t}
cd $R
for rep in 1 2; do for hm in 0 1; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --live-pmc off --opt kv_headmajor=$hm > $O/${TAG}_bench_$hm.json 2> /dev/null
python -c "
import json
d=json.loads(open('$O/${TAG}_bench_$hm.json').read().strip().splitlines()[-1]); r=d['roofline']; print('hm=$hm', d['value'], d['ms_per_step'], r['gemm_ms_per_step'], r['attention_ms_per_step'], r['other_ms_per_step'], d['config']['warm_ms_per_step'], d['config']['incremental_env_step_ms'])"
done; done
