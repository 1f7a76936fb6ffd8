cd $GRAFT_REPO_ROOT
O=gpurun_out
for prec in bf16 fp8 fp8w; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only --precision $prec --prompt-len 1024 > $O/r03_lp1024_$prec.json 2> $O/r03_lp1024_${prec}_err.txt
  python - <<PY
import json
d=json.loads(open("$O/r03_lp1024_$prec.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$prec Lp=1024: ms/step", d["ms_per_step"], "gemm_ms", r["gemm_ms_per_step"], "attn_ms", r["attention_ms_per_step"], "other_ms", r["other_ms_per_step"])
for k,v in list(r["gemm_kernels"].items())[:6]: print("   ", k, v)
PY
done
for prec in bf16 fp8; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only --precision $prec > $O/r03_lp512_$prec.json 2> $O/r03_lp512_${prec}_err.txt
  python -c "
import json
d=json.loads(open('$O/r03_lp512_$prec.json').read().strip().splitlines()[-1]); print('$prec Lp=512: ms/step', d['ms_per_step'], 'gemm_ms', d['roofline']['gemm_ms_per_step'])"
done
