#!/bin/bash
# variant binaries: bash scripts/micro/build_gemm_lab.sh r06 ppstamps -DVIMA_PP_PHASE_STAMPS; for v in FULLLINE NOSTORE1; do bash scripts/micro/build_gemm_lab.sh r06 $v -DVIMA_LAB_$v -DVIMA_PP_PHASE_STAMPS; done
# pp epilogue store shape: shipped (64-byte half lines per row and instruction) vs timing-only full 128-byte lines vs no stores; phase stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
run() { echo "== $*"; timeout 200 "$@" 2>&1 | grep -v "^$\|host fp64"; }
for v in _ppstamps _FULLLINE _NOSTORE1; do
  STAMPS=1 run $L$v 131072 2304 768 1 0 5 pp
  STAMPS=1 run $L$v 131072 3072 768 1 1 5 pp
done
VIMA_GEMM_LAB_CUS=32 STAMPS=1 run ${L}_ppstamps 32768 2304 768 1 0 3 pp
VIMA_GEMM_LAB_CUS=32 STAMPS=1 run ${L}_FULLLINE 32768 2304 768 1 0 3 pp
VIMA_GEMM_LAB_CUS=32 STAMPS=1 run ${L}_NOSTORE1 32768 2304 768 1 0 3 pp
