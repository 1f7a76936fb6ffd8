R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_all.txt 2>&1; tail -3 gpurun_out/pytest_all.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench4.txt 2>&1; tail -1 gpurun_out/bench4.txt | cut -c1-2500
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1b -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stdout.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_stdout.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_write_stdout.txt 2>&1
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_r1b
