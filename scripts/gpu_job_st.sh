cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
for st in 0 1 2 3 0 1; do
  echo "== VIMA_GEMM_ST=$st"
  VIMA_GEMM_ST=$st timeout 120 $L 65536 2304 768 1 0 7 pp | grep median
  VIMA_GEMM_ST=$st timeout 120 $L 65536 768 3072 4 0 7 pp | grep median
  VIMA_GEMM_ST=$st timeout 120 $L 65536 3072 768 1 1 7 pp | grep median
  VIMA_GEMM_ST=$st timeout 120 $L 65536 768 768 4 0 7 pp | grep median
done
