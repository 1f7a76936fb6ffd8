# Round-4 records: full GPU suite, the default bench line (side configurations, live PMC), headline / batch-1 kernel traces, PMC traffic per kernel and per
# GEMM shape (launch log), SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_v2}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; tail -3 $O/${TAG}_pytest.txt
S=$(date +%s); timeout 600 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt; E=$(date +%s)
echo "bench.py --steps 20 --warmup 5: $((E-S)) s wall" | tee -a $O/${TAG}_pytest.txt; cut -c1-250 $O/${TAG}_bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
TITLE="Round 4 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0, VIMA-200M B=256 Lp=512 bf16, 1x MI355X"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_final/bench_results.db $O/${TAG}_kernel_stats.md "$TITLE"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o bench -- python $R/bench.py --batch 1 --steps 4 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_b1/bench_results.db $O/${TAG}_b1_kernel_stats.md "Round 4 ($TAG): batch 1 cold steps, dual_stream=0, 6 steps (rocprofv3 --kernel-trace --stats -- python bench.py --batch 1 --steps 4 --warmup 2 --headline-only)"
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --live-pmc off --opt dual_stream=0 --launch-log /tmp/launches.json > /dev/null 2>&1
done
python $R/scripts/pmc_summary.py /tmp/pmc_FETCH_SIZE/bench_results.db /tmp/pmc_WRITE_SIZE/bench_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json /tmp/launches.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/sq_summary.py /tmp/pmc_sq/bench_results.db $O/${TAG}_sq_counters.md
head -20 $O/${TAG}_kernel_stats.md | cut -c1-150
