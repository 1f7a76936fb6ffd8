# secondary configurations for DESIGN.md (single GPU)
R=$GRAFT_REPO_ROOT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print(sys.argv[1], "cold ms", d["ms_per_step"], "samples/s", c["samples_per_s"], "warm ms", c["warm_ms_per_step"], "incremental ms", c.get("incremental_env_step_ms"), "gemm TF/s", r["achieved"], "step TF/s", r["whole_step_tflops"])'
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 2>&1 | tail -1 | python -c "$P" "200M B1 Lp512 Q8:"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 32 2>&1 | tail -1 | python -c "$P" "200M B32 Lp512 Q8:"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --model 20M --batch 32 --prompt-len 256 --qv 2 --words 4 2>&1 | tail -1 | python -c "$P" "20M B32 Lp256 Q4:"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --steps-history 8 2>&1 | tail -1 | python -c "$P" "200M B256 Lp512 T8:"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --prompt-len 1024 2>&1 | tail -1 | python -c "$P" "200M B256 Lp1024:"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp32 --batch 64 2>&1 | tail -1 | python -c "$P" "200M fp32 B64:"
