cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
for kb in 0 2560 1700 0 2560; do
  echo "=== NGROUP_KB=$kb"
  export VIMA_GEMM_NGROUP_KB=$kb
  timeout 120 $L 131072 3072 768 1 1 5 persist,pp | grep -E "TFLOP|differ"
  timeout 120 $L 131072 2304 768 1 0 5 persist,pp | grep -E "TFLOP|differ"
  timeout 120 $L 131072 768 3072 4 0 5 persist,pp | grep -E "TFLOP|differ"
  timeout 120 $L 81920 3072 768 1 3 5 persist,pp | grep -E "TFLOP|differ"
done
