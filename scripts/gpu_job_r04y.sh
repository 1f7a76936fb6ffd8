# A (shipped library) against B (lab build -DVIMA_LAB_NOOUT8: no fp8-copy code and no output-pointer null checks in the 256x256 epilogues), bf16 headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { # name lib opts
  VIMA_HIP_LIB=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only --live-pmc off $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('$1', j['ms_per_step'], ' '.join('%s=%.3f' % (k.split('::')[1][:22], v['ms_per_step']) for k, v in list(r['gemm_kernels'].items())[:6]), 'attn=%.3f other=%.3f' % (r['attention_ms_per_step'], r['other_ms_per_step']))"
}
for i in 1 2; do
run A $R/vima_amd/lib/libvima_hip.so ""
run B $R/vima_amd/lib/libvima_hip_noout8.so ""
done | tee $O/r04y_out8_ab.txt
