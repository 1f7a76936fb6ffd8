#!/bin/bash
# pp epilogue: shipped (LDS-transposed, 16 rows x 64 B per store instruction) vs timing-only stores straight from the MFMA layout (32 rows x 32 B, no LDS round trip)
# variant binaries: bash scripts/micro/build_gemm_lab.sh r06 ppstamps -DVIMA_PP_PHASE_STAMPS; bash scripts/micro/build_gemm_lab.sh r06 DIRECT32 -DVIMA_LAB_DIRECT32 -DVIMA_PP_PHASE_STAMPS
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
run() { echo "== $*"; timeout 200 "$@" 2>&1 | grep -v "^$\|host fp64\|differ"; }
for rep in 1 2; do
for v in _ppstamps _DIRECT32; do
  STAMPS=1 run $L$v 131072 2304 768 1 0 5 pp
  STAMPS=1 run $L$v 131072 3072 768 1 1 5 pp
  STAMPS=1 run $L$v 81920 3072 768 1 3 5 pp
done
done
