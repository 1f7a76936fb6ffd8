R=$GRAFT_REPO_ROOT
cd $R
timeout 500 python -m pytest tests/test_ops_gpu.py -q -x -k "linear" > gpurun_out/pytest7.txt 2>&1; tail -2 gpurun_out/pytest7.txt
for SH in "131072 2304 768" "131072 768 3072" "131072 768 768" "81920 3072 768"; do
python scripts/gemm_micro.py $SH 2 5 2>&1 | tail -1
done
python scripts/gemm_micro.py 131072 2304 768 1 5 2>&1 | tail -1
python scripts/gemm_micro.py 2048 768 768 0 5 2>&1 | tail -1
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench7.txt 2>&1; tail -1 gpurun_out/bench7.txt | cut -c1-1600
