# A/B of GEMM tile variants on the four big VIMA-200M shapes: bash scripts/gpu_job_gemm_ab.sh "2 5 6"
R=$GRAFT_REPO_ROOT
cd $R
TILES=${1:-"2 5 6"}
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "large_tile" 2>&1 | tail -3
for rep in 1 2; do
for T in $TILES; do
for SH in "131072 2304 768" "131072 768 3072" "81920 3072 768" "131072 768 768"; do
STAMPS=1 timeout 120 python scripts/gemm_micro.py $SH $T 5 2>&1 | tail -2
done
done
done
