#!/bin/bash
# Lab binaries of gemm_lab.hip (torch-free harness around launch_gemm): the plain build measures the SHIPPED kernels; a variant build re-creates
# round 4's timing-only / cache-policy branches on a scratch copy of gemm.hip (scripts/ablate/gemm_lab_variants.patch) and defines its macro.
#   bash scripts/micro/build_gemm_lab.sh                 -> scripts/micro/gemm_lab
#   bash scripts/micro/build_gemm_lab.sh nores -DVIMA_LAB_NORES   (also: nostore -DVIMA_LAB_NOSTORE, skew3 -DVIMA_LAB_SKEW=3, nta -DVIMA_LAB_NT_A, ntst -DVIMA_LAB_NT_ST)
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; C=$R/vima_amd/csrc
if [ -z "$1" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIMA_GEMM_LAB -I $C -o $R/scripts/micro/gemm_lab $R/scripts/micro/gemm_lab.hip
  exit 0
fi
name=$1; shift
S=$(mktemp -d); cp $C/*.h $C/*.inc $S/
patch -s -o $S/gemm.hip $C/gemm.hip $R/scripts/ablate/gemm_lab_variants.patch
sed "s#../../vima_amd/csrc/gemm.hip#$S/gemm.hip#" $R/scripts/micro/gemm_lab.hip > $S/gemm_lab.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIMA_GEMM_LAB "$@" -I $S -o $R/scripts/micro/gemm_lab_$name $S/gemm_lab.hip
