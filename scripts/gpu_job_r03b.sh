cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_error_budget_gpu.py tests/test_eval_loop.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -25 > $O/r03b_pytest.txt; cat $O/r03b_pytest.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03b_bench.json 2> $O/r03b_bench_err.txt; cut -c1-2800 $O/r03b_bench.json; tail -3 $O/r03b_bench_err.txt
