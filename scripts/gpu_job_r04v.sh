R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04v}
cd $R
OUT=$O/${TAG}_attn.txt; : > $OUT
for rep in 1 2 3; do for v in base lsum0; do
  if [ $v = base ]; then unset VIMA_HIP_LIB; else export VIMA_HIP_LIB=$R/build_ablate/libvima_hip_$v.so; fi
  echo "== $v" >> $OUT
  timeout 120 python scripts/attn_micro.py 256 12 512 64 20 2>&1 | grep "attn mode" >> $OUT
  timeout 120 python scripts/attn_micro.py 256 12 1024 64 8 2>&1 | grep "attn mode" >> $OUT
done; done
unset VIMA_HIP_LIB
cat $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" 2>&1 | tail -2
