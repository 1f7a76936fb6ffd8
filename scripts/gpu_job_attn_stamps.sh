# T5 attention microbenchmark (optionally per-phase shader-clock stamps, STAMPS=1) for experiment libraries in build_ablate/
R=$GRAFT_REPO_ROOT
cd $R
cp vima_amd/lib/libvima_hip.so /tmp/base.so
for x in "$@"; do
  [ $x = base ] || cp build_ablate/libvima_hip_$x.so vima_amd/lib/libvima_hip.so
  echo "== $x T5 512:  $(QG=1 python scripts/attn_micro.py 256 12 512 64 8 2>&1 | tail -1)"
  echo "== $x T5 1024: $(QG=1 python scripts/attn_micro.py 256 12 1024 64 4 2>&1 | tail -1)"
  echo "== $x cross D32: $(MODE=1 QG=1 python scripts/attn_micro.py 256 24 512 32 8 2>&1 | tail -1)"
  echo "   $(QG=1 CHECK=1 python scripts/attn_micro.py 16 12 500 64 1 2>&1 | grep 'max |')"
  cp /tmp/base.so vima_amd/lib/libvima_hip.so
done
