cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=${1:-r03_c}
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" > $O/${TAG}_pytest.txt; tail -4 $O/${TAG}_pytest.txt; grep -E "\[parity\]|\[fp8\]|\[fp8w\]|\[eval-loop\]|\[error-budget\]|\[stream_T\]" $O/${TAG}_pytest.txt | cut -c1-400
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -3 $O/${TAG}_smoke.txt
