# GEGLU pair over block-interleaved weights: bit-identity test, then warm / incremental / cold step with geglu_pair = 0 | 1 (one library, same box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 500 python -m pytest tests/test_policy_gpu.py -m gpu -x -q -k "geglu_pair" 2>&1 | tail -15 | tee $O/r04z_tests.txt
run() {
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --live-pmc off --opt geglu_pair=$1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['config']; s=c['secondary_cold']
print('geglu_pair=$1 cold', j['ms_per_step'], 'warm', c['warm_ms_per_step'], 'incremental', c['incremental_env_step_ms'], 'b1', s['batch_1']['ms_per_step'], 'b32', s['batch_32']['ms_per_step'])"
}
for i in 1 2; do run 0; run 1; done | tee $O/r04z_pair_ab.txt
