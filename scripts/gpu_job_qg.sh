cd $GRAFT_REPO_ROOT
for qg in 2 1; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only --opt attn_qg=$qg > gpurun_out/r03_qg$qg.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/r03_qg$qg.json').read().strip().splitlines()[-1]); r=d['roofline']
print('attn_qg=$qg ms/step', d['ms_per_step'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_ms_per_step'])"
done
