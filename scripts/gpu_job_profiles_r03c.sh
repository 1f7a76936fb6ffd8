# Round-3 closing records: bench (with CPU baseline), headline kernel trace, batch-1 / batch-32 traces.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r03_v7}
cd $R
timeout 300 python bench.py --steps 8 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt; cut -c1-300 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
TITLE="Round 3 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0, VIMA-200M B=256 Lp=512 bf16, 1x MI355X"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_final/bench_results.db $O/${TAG}_kernel_stats.md "$TITLE"
for B in 1 32; do
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o bench -- python $R/bench.py --batch $B --steps 4 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/${TAG}_prof_b${B}_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_b$B/bench_results.db $O/${TAG}_b${B}_kernel_stats.md "Round 3 ($TAG): batch $B cold steps, dual_stream=0, 6 steps (rocprofv3 --kernel-trace --stats -- python bench.py --batch $B --steps 4 --warmup 2 --headline-only)"
done
