cd $GRAFT_REPO_ROOT
for qg in 1 2; do QG=$qg timeout 300 python scripts/attn_micro.py 256 12 1024 64 5 2>&1 | tail -1; done
for qg in 1 2; do MODE=1 QG=$qg timeout 300 python scripts/attn_micro.py 256 24 512 32 5 2>&1 | tail -1; done
timeout 1200 python -m pytest tests/test_policy_gpu.py tests/test_eval_loop.py tests/test_fp8_gpu.py tests/test_baselines_gpu.py -m gpu -q -x 2>&1 | tail -3
