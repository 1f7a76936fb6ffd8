R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04o}
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "full_load or pingpong or wide_tile or epilogues or residual_stream" > $O/${TAG}_pytest.txt 2>&1; tail -5 $O/${TAG}_pytest.txt
python -c "
import torch, bench
print('numa', bench.gpu_numa_cpus(0))
"
