R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "linear" 2>&1 | tail -3
for rep in 1 2; do
for BO in 0 1; do
for P in 1 0; do
for SH in "131072 768 3072" "131072 2304 768" "81920 3072 768"; do
  BF16OUT=$BO VIMA_GEMM_PERSIST=$P STAMPS=1 timeout 120 python scripts/gemm_micro.py $SH 0 5 2>&1 | tail -3 | grep -v "real time" | cut -c1-200
done
done
done
done
