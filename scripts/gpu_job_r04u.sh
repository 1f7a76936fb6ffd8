R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04u}
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/${TAG}_pytest.txt 2>&1; tail -4 $O/${TAG}_pytest.txt
CHECK=1 timeout 120 python scripts/attn_micro.py 64 12 512 64 3 2>&1 | grep "max"
for rep in 1 2; do
timeout 120 python scripts/attn_micro.py 256 12 512 64 20 2>&1 | grep "attn mode"
timeout 120 python scripts/attn_micro.py 256 12 1024 64 8 2>&1 | grep "attn mode"
done
timeout 900 python -m pytest tests/test_policy_gpu.py -x -q -k "golden or headline or error" > $O/${TAG}_pytest2.txt 2>&1; tail -3 $O/${TAG}_pytest2.txt
