# Round-3 final deliverables on the GPU box: smoke, bench (with CPU baseline), rocprofv3 kernel trace and PMC passes, batch-1 / batch-32 traces.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r03_v5}
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -2 $O/${TAG}_smoke.txt
timeout 600 python bench.py --steps 8 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt; cut -c1-1500 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
TITLE="Round 3 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0, VIMA-200M B=256 Lp=512 bf16, 1x MI355X"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_final/bench_results.db $O/${TAG}_kernel_stats.md "$TITLE"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --opt dual_stream=0 > /dev/null 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc_fetch/bench_results.db /tmp/pmc_write/bench_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json
for B in 1 32; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o bench -- python $R/bench.py --batch $B --steps 4 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_b${B}_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_b$B/bench_results.db $O/${TAG}_b${B}_kernel_stats.md "Round 3 ($TAG): batch $B cold steps, dual_stream=0, 6 steps (rocprofv3 --kernel-trace --stats -- python bench.py --batch $B --steps 4 --warmup 2 --headline-only)"
done
ls $O | grep $TAG
