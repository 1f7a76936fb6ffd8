#!/bin/bash
# Experiment builds of attention.hip (DESIGN.md section 4, T5 attention): one libvima_hip.so per setting of the kernel's experiment switches
# (VIMA_ATTN_ABL bits: timing only, wrong results; VIMA_ATTN_NSTG1 / VIMA_ATTN_OCC1 / VIMA_ATTN_V2...: real variants) -> build_ablate/libvima_hip_<tag>.so.
#   usage: scripts/build_attn_variants.sh tag1:"-DFLAG=..." tag2:"..."        run one with VIMA_HIP_LIB=build_ablate/libvima_hip_<tag>.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/vima_amd/csrc
mkdir -p $R/build_ablate
bash $C/build.sh > /dev/null
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -c $C/attention.hip -o $R/build_ablate/attention_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ablate/libvima_hip_$tag.so $C/obj/gemm.o $C/obj/elementwise.o $R/build_ablate/attention_$tag.o $C/obj/vima_api.o $C/obj/comm.o $C/obj/preprocess.o $C/obj/baseline_kernels.o -ldl && echo "built $tag" ) &
done
wait
rm -f $R/build_ablate/attention_*.o
ls $R/build_ablate
