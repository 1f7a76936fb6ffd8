# gate-epilogue multiplier rows requested ahead (ring tiles): targeted tests, then previous library against this one (warm / incremental / cold), same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 60 python -m pytest tests/test_ops_gpu.py tests/test_policy_gpu.py -m gpu -x -q -k "linear_epilogues or resident_kernel or geglu or golden" 2>&1 | tail -3 | tee $O/r04zz_tests.txt
run() {
  VIMA_HIP_LIB=$2 timeout 60 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --live-pmc off 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['config']; s=c['secondary_cold']
print('$1 cold', j['ms_per_step'], 'warm', c['warm_ms_per_step'], 'incremental', c['incremental_env_step_ms'], 'b1', s['batch_1']['ms_per_step'], 'b32', s['batch_32']['ms_per_step'])"
}
run prev $R/vima_amd/lib/libvima_hip_prev.so | tee $O/r04zz_ab.txt
run new $R/vima_amd/lib/libvima_hip.so | tee -a $O/r04zz_ab.txt
