# Quick GPU-box check: full -m gpu suite (with the [parity] report lines), smoke(), default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r02_a}
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" > $O/${TAG}_pytest.txt; tail -5 $O/${TAG}_pytest.txt; grep "\[parity\]" $O/${TAG}_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -3 $O/${TAG}_smoke.txt
timeout 900 python bench.py --steps 8 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt; cut -c1-3000 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench_stderr.txt
