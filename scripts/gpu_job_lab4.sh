cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
export STAMPS=1
timeout 120 $L 8192 8192 8192 1 0 3 | grep -v "host fp64"
timeout 120 $L 131072 2304 768 1 0 3 | grep -v "host fp64"
for sk in 0 2000 4000 8000; do
  echo "=== SKEW=$sk"
  SKEW=$sk timeout 120 $L 131072 2304 768 1 0 3 pp | grep -v "host fp64"
  SKEW=$sk timeout 120 $L 131072 768 768 4 0 3 pp
  SKEW=$sk timeout 120 $L 131072 768 3072 4 0 3 pp
done
