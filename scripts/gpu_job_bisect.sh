cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_policy_gpu.py tests/test_eval_loop.py tests/test_fp8_gpu.py tests/test_fp8w_gpu.py tests/test_baselines_gpu.py -m gpu -q -x 2>&1 | tail -3
