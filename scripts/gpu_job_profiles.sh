# Round deliverables on the GPU box: full -m gpu suite, bench (with CPU baseline), rocprofv3 kernel trace and PMC passes.
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_all.txt 2>&1; tail -3 gpurun_out/pytest_all.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_final.txt 2>&1; tail -1 gpurun_out/bench_final.txt | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --opt dual_stream=0 > $R/gpurun_out/prof_stdout.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --opt dual_stream=0 > $R/gpurun_out/pmc_fetch_stdout.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --opt dual_stream=0 > $R/gpurun_out/pmc_write_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --opt dual_stream=0 > $R/gpurun_out/pmc_sq_stdout.txt 2>&1
ls $R/gpurun_out/prof_final $R/gpurun_out/pmc_sq
