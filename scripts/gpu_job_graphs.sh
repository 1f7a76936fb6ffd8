cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(sys.argv[1], "cold ms", d["ms_per_step"], "warm ms", c["warm_ms_per_step"], "incremental ms", c.get("incremental_env_step_ms"))'
for G in 0 1; do
for B in 1 32 256; do
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --batch $B --opt graphs=$G 2>&1 | tail -1 | python -c "$P" "200M B$B graphs=$G:"
done
done
