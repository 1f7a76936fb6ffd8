cd $GRAFT_REPO_ROOT
timeout 600 python scripts/gemm_midm.py 2304 2>&1 | grep "^M"
