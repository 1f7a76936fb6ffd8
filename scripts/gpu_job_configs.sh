# secondary configurations for DESIGN.md (single GPU): batch sweep at VIMA-200M and the VIMA-20M config of BASELINE.json
R=$GRAFT_REPO_ROOT
cd $R
for B in 1 32; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('200M batch', $B, 'cold ms', d['ms_per_step'], 'samples/s', c['samples_per_s'], 'warm ms', c['warm_ms_per_step'], 'gemm TF/s', r['achieved'], 'step TF/s', r['whole_step_tflops'])"
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --model 20M --batch 32 --prompt-len 256 --qv 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('20M batch 32 Lp256 Q4: cold ms', d['ms_per_step'], 'samples/s', c['samples_per_s'], 'warm ms', c['warm_ms_per_step'], 'gemm TF/s', r['achieved'], 'step TF/s', r['whole_step_tflops'])"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp32 --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('200M fp32-operand mode batch 64: cold ms', d['ms_per_step'], 'samples/s', c['samples_per_s'], 'gemm TF/s', r['achieved'])"
