cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py tests/test_fp8_gpu.py -m gpu -q -x -k "norm or vit or golden or headline or invarian or fp8 or obj or chunk" 2>&1 | tail -4
for cfg in "1 1" "0 0"; do
  set -- $cfg
  VIMA_LN_ROWS2=$1 VIMA_VIT_ATTN_LDS=$2 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only > gpurun_out/r03_tail_bench_$1$2.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/r03_tail_bench_$1$2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ln_rows2=$1 vit_lds=$2 ms/step', d['ms_per_step'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_ms_per_step'])"
done
