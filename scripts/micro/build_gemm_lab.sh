#!/bin/bash
# Lab binaries of gemm_lab.hip (torch-free harness around launch_gemm): the plain build measures the SHIPPED kernels; a variant build re-creates
# timing-only / experiment branches on a scratch copy of the sources and defines its macro. The shipped kernels carry none of these switches.
#   bash scripts/micro/build_gemm_lab.sh                 -> scripts/micro/gemm_lab
#   round 4 (scripts/ablate/gemm_lab_variants.patch, applies to the round-5 tree; kept for the record):
#     bash scripts/micro/build_gemm_lab.sh nores -DVIMA_LAB_NORES   (also: nostore -DVIMA_LAB_NOSTORE, skew3 -DVIMA_LAB_SKEW=3, nta -DVIMA_LAB_NT_A, ntst -DVIMA_LAB_NT_ST)
#   round 6 (scripts/ablate/gemm_lab_r06.patch over gemm.hip + gemm_q4.inc):
#     bash scripts/micro/build_gemm_lab.sh r06 <name> <flags>, flags among
#       -DVIMA_Q4_NOWAIT | _NOBAR | _NODMA | _NOREAD   gemm_q4_kernel main loop without its counted DMA wait / phase barriers / LDS-DMA requests / fragment reads (timing only)
#       -DVIMA_Q4_KT_STAMPS                            per-K-tile clock stamps of gemm_q4_kernel (with STAMPS=1 KTSTAMPS=1)
#       -DVIMA_PP_PHASE_STAMPS                         per-phase stamps of gemm_pp_kernel (shipped source, macro only)
#       -DVIMA_LAB_AUXD=n                              prefetch distance of the stream / gate epilogue's per-row operand (shipped: 3)
#       -DVIMA_LAB_FULLLINE | _NOSTORE1 | _DIRECT32    EPI 1 stores as full 128-byte lines / no stores / straight from the MFMA layout in 32-byte pieces (timing only)
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; C=$R/vima_amd/csrc
if [ -z "$1" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIMA_GEMM_LAB -I $C -o $R/scripts/micro/gemm_lab $R/scripts/micro/gemm_lab.hip
  exit 0
fi
S=$(mktemp -d); cp $C/*.h $C/*.inc $C/gemm.hip $S/
if [ "$1" = r06 ]; then
  shift; name=$1; shift
  (cd $S && patch -s -p1 < $R/scripts/ablate/gemm_lab_r06.patch)
else
  name=$1; shift
  patch -s -o $S/gemm.hip $C/gemm.hip $R/scripts/ablate/gemm_lab_variants.patch
fi
sed "s#../../vima_amd/csrc/gemm.hip#$S/gemm.hip#" $R/scripts/micro/gemm_lab.hip > $S/gemm_lab.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIMA_GEMM_LAB "$@" -I $S -o $R/scripts/micro/gemm_lab_$name $S/gemm_lab.hip
