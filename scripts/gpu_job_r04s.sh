R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04s}
cd $R
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "head_major or kv_cache or incremental or restart" > $O/${TAG}_pytest.txt 2>&1; tail -15 $O/${TAG}_pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --live-pmc off > $O/${TAG}_bench.json 2> /dev/null
python -c "
import json
d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['gemm_ms_per_step'], r['attention_ms_per_step'], r['other_ms_per_step'], d['config']['warm_ms_per_step'], d['config']['incremental_env_step_ms'])"
