# kernel-trace summary of the headline step (single stream) -> gpurun_out/${TAG}_kernel_stats.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03_a}; shift
cd /tmp && export TMPDIR=/tmp
TITLE="Round 3 ($TAG): rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 $*, VIMA-200M B=256 Lp=512 bf16, 1x MI355X"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 "$@" > $O/${TAG}_prof_stdout.txt 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_$TAG/bench_results.db $O/${TAG}_kernel_stats.md "$TITLE"
