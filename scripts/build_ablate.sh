#!/bin/bash
# Experiment builds behind DESIGN.md 4.2 (timing only, WRONG RESULTS): one libvima_hip.so per GEMM main-loop ablation
# -> build_ablate/. The ablation branches are not in the product source: scripts/ablate/*.patch re-creates them on a
# scratch copy of gemm.hip / attention.hip (generated against the source of the commit that added this script; if a
# patch no longer applies, re-derive it by hand -- it only removes one instruction class per bit from the main loop).
#   VIMA_GEMM_ABLATE bits: 1 = no LDS-DMA in the main loop, 2 = no s_barrier, 4 = no fragment ds_reads, 8 = no vmcnt wait, 16 = no MFMAs
#   VIMA_ATTN_ABLATE bits: 1 = no per-tile barrier, 2 = no softmax exp2;  VIMA_ATTN_OCC = waves per SIMD at D=64
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/vima_amd/csrc
S=$(mktemp -d)
mkdir -p $R/build_ablate
bash $C/build.sh > /dev/null
cp $C/*.h $S/ && mkdir -p $S/../../include 2>/dev/null || true
patch -s -o $S/gemm.hip $C/gemm.hip $R/scripts/ablate/gemm_ablate.patch
sed -i 's#"kernels.h"#"'$C'/kernels.h"#' $S/gemm.hip
for x in ${@:-1 2 4 8 3 5 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVIMA_GEMM_ABLATE=$x -c $S/gemm.hip -o $S/gemm_abl$x.o &
done
wait
for x in ${@:-1 2 4 8 3 5 7 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ablate/libvima_hip_abl$x.so $S/gemm_abl$x.o $C/obj/elementwise.o $C/obj/attention.o $C/obj/vima_api.o $C/obj/comm.o $C/obj/preprocess.o $C/obj/baseline_kernels.o -ldl
done
ls -la $R/build_ablate
