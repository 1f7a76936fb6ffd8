# Round 3: fast bf16 GELU + resident tile rule: op / policy tests that gate them, micro timings, step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03u}
cd $R
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -k "resident or small_tiles or epilogues or linear" > $O/${TAG}_pytest_ops.txt 2>&1; tail -8 $O/${TAG}_pytest_ops.txt
timeout 500 python -m pytest tests/test_policy_gpu.py -x -q -k "resident or matches_reference_golden or incremental_decoding_matches" > $O/${TAG}_pytest_policy.txt 2>&1; tail -8 $O/${TAG}_pytest_policy.txt
timeout 420 python scripts/small_m_ab.py micro 20 > $O/${TAG}_small_m_ab.txt 2>&1; cat $O/${TAG}_small_m_ab.txt | tail -40
