cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
export STAMPS=1
timeout 120 $L 131072 2304 768 1 0 5 persist,pp | grep -v "host fp64"
timeout 120 $L 131072 3072 768 1 1 5 persist,pp
BIAS=1 timeout 120 $L 81920 3072 768 1 3 5 persist,pp
timeout 120 $L 131072 768 768 4 0 5 persist,pp
timeout 120 $L 131072 768 3072 4 0 5 persist,pp
