#!/bin/bash
# Build libvima_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU present.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result"
pids=()
for f in gemm elementwise attention vima_api comm preprocess baseline_kernels; do
  src="$HERE/$f.hip"; obj="$HERE/obj/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/kernels.h" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/../../include/vima_hip.h" -nt "$obj" ] \
     || { [ "$f" = vima_api ] && [ "$HERE/baselines.inc" -nt "$obj" ]; } || { [ "$f" = gemm ] && [ "$HERE/gemm_small.inc" -nt "$obj" ]; }; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvima_hip.so" "$HERE"/obj/gemm.o "$HERE"/obj/elementwise.o "$HERE"/obj/attention.o "$HERE"/obj/vima_api.o "$HERE"/obj/comm.o "$HERE"/obj/preprocess.o "$HERE"/obj/baseline_kernels.o -ldl
echo "built $OUT/libvima_hip.so"
