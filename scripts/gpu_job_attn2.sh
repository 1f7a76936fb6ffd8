cd $GRAFT_REPO_ROOT
for qg in 2 1; do QG=$qg timeout 300 python scripts/attn_micro.py 256 12 512 64 10 2>&1 | tail -1; done
QG=2 timeout 300 python scripts/attn_micro.py 256 12 1024 64 5 2>&1 | tail -1
QG=2 CHECK=1 timeout 300 python scripts/attn_micro.py 16 12 500 64 1 2>&1 | tail -2 | head -1
MODE=1 QG=2 timeout 300 python scripts/attn_micro.py 256 24 512 32 5 2>&1 | tail -1
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or attn" 2>&1 | tail -3
