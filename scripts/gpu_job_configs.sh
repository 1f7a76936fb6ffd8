# Secondary configurations of BASELINE.json on one GPU (records for profiles/; the headline line is bench.py's default)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_secondary_configs.txt
: > $O
run() { echo "== $*" >> $O; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'dtype': d['dtype'], 'workload': d['config']['workload'], 'samples_per_s': d['config']['samples_per_s'], 'all_gemm_tflops': r['all_gemm']['achieved'], 'whole_step_tflops': r['whole_step_tflops'], 'attention_ms': r['attention_ms_per_step'], 'other_ms': r['other_ms_per_step']}))" >> $O; }
run --model 20M --batch 32 --prompt-len 256 --qv 2 --words 4                       # configs[1]
run --prompt-len 1024                                                              # configs[4] shape, bf16
run --prompt-len 1024 --precision fp8w                                             # configs[4]: fp8 weights
run --precision fp8w                                                               # headline shape, fp8 weights
run --steps-history 8                                                              # T = 8 history re-fed like the reference
run --batch 64 --precision fp32                                                    # fp32-operand parity mode
cat $O
