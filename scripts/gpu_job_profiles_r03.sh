# Round-3 deliverables on the GPU box: bench (with CPU baseline), rocprofv3 kernel trace and PMC passes, fp8 kernel trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r03_v2}
cd $R
timeout 600 python bench.py --steps 8 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt; cut -c1-1200 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
TITLE="Round 3 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0, VIMA-200M B=256 Lp=512 bf16, 1x MI355X"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_final/bench_results.db $O/${TAG}_kernel_stats.md "$TITLE"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc_fetch/bench_results.db /tmp/pmc_write/bench_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/sq_summary.py /tmp/pmc_sq/bench_results.db $O/${TAG}_sq_counters.md
TITLE8="Round 3 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 --precision fp8 --prompt-len 1024 (BASELINE configs[4] shape), VIMA-200M B=256, 1x MI355X"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fp8 -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 --precision fp8 --prompt-len 1024 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_fp8/bench_results.db $O/${TAG}_fp8_lp1024_kernel_stats.md "$TITLE8"
ls $O | grep $TAG
