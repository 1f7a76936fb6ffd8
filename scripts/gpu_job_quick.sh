cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "layernorm or lds_dma" 2>&1 | tail -4
