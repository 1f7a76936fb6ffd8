cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
export STAMPS=1
for abl in 0 1 2 4 6 8 14; do
  echo "=== ABL=$abl"
  ABL=$abl timeout 120 $L 8192 8192 8192 1 0 2 pp | grep -v "host fp64"
  ABL=$abl timeout 120 $L 131072 2304 768 1 0 2 pp | grep -v "host fp64"
done
