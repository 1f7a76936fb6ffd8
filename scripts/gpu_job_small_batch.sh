R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for B in 32 1; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o bench -- python $R/bench.py --batch $B --steps 4 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 > $O/prof_b${B}_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_b$B/bench_results.db $O/r02_b${B}_kernel_stats.md "Round 2: batch $B, dual_stream=0, 6 steps"
head -40 $O/r02_b${B}_kernel_stats.md
done
