cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py -m gpu -q -x -k "attention or attn or golden or step or increment or norm" 2>&1 | tail -3
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_split_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_split_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step', d['ms_per_step'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_ms_per_step'])
for k in ('warm_ms_per_step','incremental_ms_per_step','secondary_cold'):
    print(k, json.dumps(d.get(k) if k in d else d.get('config',{}).get(k))[:400])
print([k for k in d.keys()])
PY
