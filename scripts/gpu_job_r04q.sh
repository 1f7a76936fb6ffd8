R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04q}
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/${TAG}_pytest.txt 2>&1; tail -4 $O/${TAG}_pytest.txt
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or t5" > $O/${TAG}_pytest2.txt 2>&1; tail -3 $O/${TAG}_pytest2.txt
timeout 300 python bench.py --steps 10 --warmup 3 --headline-only --no-cpu-baseline > $O/${TAG}_bench.json 2> /dev/null
timeout 300 python bench.py --steps 5 --warmup 2 --headline-only --no-cpu-baseline --prompt-len 1024 > $O/${TAG}_bench1024.json 2> /dev/null
python -c "
import json
for f in ('bench','bench1024'):
    d=json.loads(open('$O/${TAG}_'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['gemm_ms_per_step'], r['attention_ms_per_step'], r['other_ms_per_step'])"
