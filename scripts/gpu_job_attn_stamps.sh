# T5 attention microbenchmark (optionally per-phase shader-clock stamps, STAMPS=1) for experiment libraries in build_ablate/
R=$GRAFT_REPO_ROOT
cd $R
cp vima_amd/lib/libvima_hip.so /tmp/base.so
for x in "$@"; do
  [ $x = base ] || cp build_ablate/libvima_hip_$x.so vima_amd/lib/libvima_hip.so
  for qg in 1 2; do
    echo "== $x QG=$qg"; QG=$qg python scripts/attn_micro.py 256 12 512 64 8 2>&1 | tail -2
  done
  QG=2 CHECK=1 python scripts/attn_micro.py 16 12 500 64 1 2>&1 | tail -2 | head -1
  cp /tmp/base.so vima_amd/lib/libvima_hip.so
done
