# Round 4, job d: full GPU suite + headline + kernel trace (vit_attn_lds swizzle)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04d}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; tail -6 $O/${TAG}_pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --headline-only --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt
python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['gemm_ms_per_step'], r['attention_ms_per_step'], r['other_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_d/bench_results.db $O/${TAG}_kernel_stats.md "Round 4 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0"
grep "vit_attn\|attn_mfma4\|layernorm_rows2\|attn_split" $O/${TAG}_kernel_stats.md | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/sq_summary.py /tmp/pmc_sq/bench_results.db $O/${TAG}_sq_counters.md; grep "vit_attn\|attn_split\|attn_mfma4" $O/${TAG}_sq_counters.md | cut -c1-200
