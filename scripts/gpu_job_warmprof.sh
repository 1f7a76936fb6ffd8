R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_warm -o warm -- python $R/scripts/warm_steps.py warm 10 256 > $O/r03_warm_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_warm/warm_results.db $O/r03_warm_kernel_stats.md "Round 3: 12 WARM steps (prompt cached) B=256, VIMA-200M Lp=512 bf16 + one prompt assembly"
tail -1 $O/r03_warm_stdout.txt
