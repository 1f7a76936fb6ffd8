cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
for kb in 2560 0 1024 1536 4096 2560; do
  echo "== VIMA_GEMM_NGROUP_KB=$kb"
  VIMA_GEMM_NGROUP_KB=$kb timeout 120 $L 65536 2304 768 1 0 7 pp | grep median
  VIMA_GEMM_NGROUP_KB=$kb timeout 120 $L 65536 3072 768 1 1 7 pp | grep median
  VIMA_GEMM_NGROUP_KB=$kb timeout 120 $L 65536 768 768 4 0 7 pp | grep median
done
