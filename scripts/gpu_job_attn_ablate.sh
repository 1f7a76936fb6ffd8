# T5 attention kernel ablations (timing only): swaps experiment libraries in on the GPU box copy
R=$GRAFT_REPO_ROOT
cd $R
cp vima_amd/lib/libvima_hip.so /tmp/base.so
for x in base $@; do
  if [ $x = base ]; then cp /tmp/base.so vima_amd/lib/libvima_hip.so; else cp build_ablate/libvima_hip_$x.so vima_amd/lib/libvima_hip.so; fi
  echo "== $x: $(python scripts/attn_micro.py 256 12 512 64 3 | tail -1)"
done
cp /tmp/base.so vima_amd/lib/libvima_hip.so
