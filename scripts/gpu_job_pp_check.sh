# ping-pong GEMM in the product: new bit-identity test, option matrix, headline bench with gemm_pp 0 / 1
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pingpong or wide_tile or linear" 2>&1 | tail -5 > $O/r03_pp_pytest.txt
timeout 900 python -m pytest tests/test_policy_gpu.py -m gpu -q -x -k "every_option or benchmarked_configs" 2>&1 | tail -5 >> $O/r03_pp_pytest.txt
cat $O/r03_pp_pytest.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only --opt gemm_pp=0 > $O/r03_pp0_bench.json 2> $O/r03_pp0_err.txt; cut -c1-600 $O/r03_pp0_bench.json
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --headline-only --opt gemm_pp=1 > $O/r03_pp1_bench.json 2> $O/r03_pp1_err.txt; cut -c1-600 $O/r03_pp1_bench.json
