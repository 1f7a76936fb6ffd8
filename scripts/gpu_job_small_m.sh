# Round 3, small-M path: bitwise tests of gemm_resident_kernel, per-shape micro timings, step A/B (ring tiles vs resident kernel), stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03s}
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "resident or small_tiles" > $O/${TAG}_pytest_ops.txt 2>&1; tail -15 $O/${TAG}_pytest_ops.txt
timeout 400 python -m pytest tests/test_policy_gpu.py -x -q -k "resident or incremental_decoding_matches or dual_stream_and_pruning" > $O/${TAG}_pytest_policy.txt 2>&1; tail -15 $O/${TAG}_pytest_policy.txt
timeout 200 python scripts/small_m_stamps.py > $O/${TAG}_stamps.txt 2>&1; cat $O/${TAG}_stamps.txt
HIP_FORCE_DEV_KERNARG=1 timeout 200 python scripts/small_m_stamps.py > $O/${TAG}_stamps_devkernarg1.txt 2>&1; grep "^M" $O/${TAG}_stamps_devkernarg1.txt | cut -c1-150
HIP_FORCE_DEV_KERNARG=0 timeout 200 python scripts/small_m_stamps.py > $O/${TAG}_stamps_devkernarg0.txt 2>&1; grep "^M" $O/${TAG}_stamps_devkernarg0.txt | cut -c1-150
timeout 420 python scripts/small_m_ab.py micro 20 > $O/${TAG}_small_m_ab.txt 2>&1; cat $O/${TAG}_small_m_ab.txt | tail -60
