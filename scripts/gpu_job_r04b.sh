# Round 4, job b: fp8 hardening tests, 8-rank launcher test, the full default bench line (side configurations + live PMC) with its wall time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04b}
cd $R
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_parallel_gpu.py -x -q > $O/${TAG}_pytest.txt 2>&1; tail -15 $O/${TAG}_pytest.txt
S=$(date +%s)
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench_stderr.txt
E=$(date +%s); echo "bench.py default run: $((E-S)) s wall" | tee $O/${TAG}_bench_wall.txt
tail -5 $O/${TAG}_bench_stderr.txt
python - <<PY
import json
d = json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d["config"]["secondary_cold"], indent=1)[:4000])
r = d["roofline"]
print(r["traffic"], r["traffic_source"])
print(json.dumps(r.get("traffic_per_shape"), indent=0)[:3000])
PY
