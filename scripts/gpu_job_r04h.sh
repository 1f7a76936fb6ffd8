R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04h}
cd $R
OUT=$O/${TAG}_attn.txt; : > $OUT
for rep in 1 2; do
for cfg in "base 1" "base 2" "qg2s2 2"; do
  set -- $cfg; v=$1; export QG=$2
  if [ $v = base ]; then unset VIMA_HIP_LIB; else export VIMA_HIP_LIB=$R/build_ablate/libvima_hip_$v.so; fi
  echo "== $v QG=$QG" >> $OUT
  timeout 120 python scripts/attn_micro.py 256 12 512 64 20 2>&1 | grep "attn mode" >> $OUT
  timeout 120 python scripts/attn_micro.py 256 12 1024 64 8 2>&1 | grep "attn mode" >> $OUT
done; done
unset VIMA_HIP_LIB
cat $OUT
