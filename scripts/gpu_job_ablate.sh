# GEMM main-loop ablations (timing only): swaps the experiment libraries in on the GPU box copy
R=$GRAFT_REPO_ROOT
cd $R
cp vima_amd/lib/libvima_hip.so /tmp/base.so
for x in base $@; do
  if [ $x = base ]; then cp /tmp/base.so vima_amd/lib/libvima_hip.so; else cp build_ablate/libvima_hip_$x.so vima_amd/lib/libvima_hip.so; fi
  echo "== ablate $x"
  for SH in "131072 768 3072" "131072 2304 768"; do
    STAMPS=1 timeout 120 python scripts/gemm_micro.py $SH 2 5 2>&1 | tail -3 | cut -c1-200
  done
done
cp /tmp/base.so vima_amd/lib/libvima_hip.so
