# Round 4, job f: persistent-over-query-blocks T5 attention: correctness vs the exact generic kernel, op tests, timing A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04f}
cd $R
OUT=$O/${TAG}_attn.txt; : > $OUT
for v in base $2; do
  if [ $v = base ]; then unset VIMA_HIP_LIB; else export VIMA_HIP_LIB=$R/build_ablate/libvima_hip_$v.so; fi
  echo "== $v" >> $OUT
  CHECK=1 timeout 120 python scripts/attn_micro.py 256 12 512 64 10 2>&1 | grep "attn mode\|max" >> $OUT
  timeout 120 python scripts/attn_micro.py 256 12 1024 64 5 2>&1 | grep "attn mode" >> $OUT
  MODE=1 LQ=576 timeout 120 python scripts/attn_micro.py 64 24 512 32 5 2>&1 | grep "attn mode" >> $OUT
done
unset VIMA_HIP_LIB
cat $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/${TAG}_pytest.txt 2>&1; tail -5 $O/${TAG}_pytest.txt
