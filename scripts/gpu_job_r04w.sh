# A/B on one box: does the head-major branch in the bf16-output epilogue cost the row-major launches anything?
# A = shipped library, B = same sources with -DVIMA_LAB_NOHM (branch compiled out; kv_headmajor=0), C = shipped library with kv_headmajor=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { # name lib opts
  VIMA_HIP_LIB=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only --live-pmc off $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('$1', j['ms_per_step'], ' '.join('%s=%.3f' % (k.split('::')[1][:22], v['ms_per_step']) for k, v in list(r['gemm_kernels'].items())[:5]), 'attn=%.3f other=%.3f' % (r['attention_ms_per_step'], r['other_ms_per_step']))"
}
for i in 1 2; do
run A $R/vima_amd/lib/libvima_hip.so ""
run B $R/vima_amd/lib/libvima_hip_nohm.so "--opt kv_headmajor=0"
run C $R/vima_amd/lib/libvima_hip.so "--opt kv_headmajor=0"
done | tee $O/r04w_hm_ab.txt
