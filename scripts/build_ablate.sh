#!/bin/bash
# experiment builds: one libvima_hip.so per GEMM main-loop ablation (wrong results, timing only) -> build_ablate/
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/vima_amd/csrc
mkdir -p $R/build_ablate
bash $C/build.sh > /dev/null
for x in ${@:-1 2 4 8 3 5 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVIMA_GEMM_ABLATE=$x -c $C/gemm.hip -o /tmp/gemm_abl$x.o &
done
wait
for x in ${@:-1 2 4 8 3 5 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ablate/libvima_hip_abl$x.so /tmp/gemm_abl$x.o $C/obj/elementwise.o $C/obj/attention.o $C/obj/vima_api.o
done
ls -la $R/build_ablate
