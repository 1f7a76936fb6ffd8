cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py -m gpu -q -x -k "attention or attn or golden or headline" 2>&1 | tail -4
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only > gpurun_out/r03_attn_bench.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03_attn_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step', d['ms_per_step'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_ms_per_step'])"
