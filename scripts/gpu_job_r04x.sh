# head-major output as its own instantiation (EPI 5): tests, then A (shipped library) against B (lab build without any head-major code, kv_headmajor=0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 400 python -m pytest tests/test_policy_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "head_major or persistent_epilogue or restart or gemm" 2>&1 | tail -3 | tee $O/r04x_tests.txt
run() { # name lib opts
  VIMA_HIP_LIB=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only --live-pmc off $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('$1', j['ms_per_step'], ' '.join('%s=%.3f' % (k.split('::')[1][:22], v['ms_per_step']) for k, v in list(r['gemm_kernels'].items())[:6]), 'attn=%.3f other=%.3f' % (r['attention_ms_per_step'], r['other_ms_per_step']))"
}
for i in 1 2; do
run A $R/vima_amd/lib/libvima_hip.so ""
run B $R/vima_amd/lib/libvima_hip_nohm.so "--opt kv_headmajor=0"
done | tee $O/r04x_hm_ab.txt
