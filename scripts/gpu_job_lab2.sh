cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
export STAMPS=1
timeout 120 $L 8192 8192 8192 1 0 3 pp
timeout 120 $L 131072 2304 768 1 0 3 pp
timeout 120 $L 131072 768 3072 4 0 3 pp
timeout 120 $L 131072 768 768 4 0 3 pp
