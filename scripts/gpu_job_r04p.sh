R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04p}
cd $R
OUT=$O/${TAG}_lab.txt; : > $OUT
NORES3=1 timeout 100 scripts/micro/gemm_lab_base 65536 768 768 3 0 7 tile,persist,pp >> $OUT 2>&1
BIAS=1 NORES3=1 timeout 100 scripts/micro/gemm_lab_base 65536 768 768 3 0 7 tile,persist,pp >> $OUT 2>&1
timeout 100 scripts/micro/gemm_lab_base 65536 768 768 3 0 5 tile,pp >> $OUT 2>&1
grep -v "clocks per\|stamps" $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "full_load or large_tile or pingpong or epilogues" > $O/${TAG}_pytest.txt 2>&1; tail -3 $O/${TAG}_pytest.txt
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or exact or headline" > $O/${TAG}_pytest2.txt 2>&1; tail -3 $O/${TAG}_pytest2.txt
timeout 300 python bench.py --steps 10 --warmup 3 --headline-only --no-cpu-baseline > $O/${TAG}_bench.json 2> /dev/null
python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['gemm_ms_per_step'])
for k,v in r['gemm_kernels'].items(): print(k, v['ms_per_step'], v['launches'], v['tflops'])"
