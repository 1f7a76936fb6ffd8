cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py -m gpu -q -x -k "attention or attn or knob or option or exact or golden" 2>&1 | tail -3
