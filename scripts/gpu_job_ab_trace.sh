# per-kernel average durations for two option settings (kernel-trace only): bash scripts/gpu_job_ab_trace.sh "opt=a" "opt=b"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats -d /tmp/ab$i -o ab -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --opt dual_stream=0 --opt $O > /tmp/ab$i.log 2>&1
  DB=$(find /tmp/ab$i -name "*.db" | head -1)
  python $R/scripts/rocprof_summary.py $DB $R/gpurun_out/ab_$i.md "$O" > /dev/null
  echo "== $O"; sed -n 7,22p $R/gpurun_out/ab_$i.md | cut -c1-120
done
