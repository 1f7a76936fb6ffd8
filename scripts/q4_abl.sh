#!/bin/bash
# variant binaries: for v in NOWAIT NOBAR NODMA NOREAD KT_STAMPS; do bash scripts/micro/build_gemm_lab.sh r06 q4$v -DVIMA_Q4_$v; done; bash scripts/micro/build_gemm_lab.sh r06 ppstamps -DVIMA_PP_PHASE_STAMPS
# gemm_q4_kernel: difference map against gemm_pp_kernel on small problems + timing-only ablations of its main loop (lab builds -DVIMA_Q4_NOWAIT / _NOBAR / _NODMA / _NOREAD)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
run() { echo "== $*"; timeout 200 "$@" 2>&1 | grep -v "^$"; }
export DIFFMAP=1
run $L 8192 3072 768 1 1 2 pp,q4
run $L 16384 768 768 4 0 2 pp,q4
run $L 16384 768 768 1 3 2 pp,q4
SSQ=1 BIAS=1 run $L 16384 768 3072 4 0 2 pp,q4
unset DIFFMAP
STAMPS=1 KTSTAMPS=1 run ${L}_q4KT_STAMPS 131072 2304 768 1 0 3 q4
STAMPS=1 KTSTAMPS=1 run ${L}_q4KT_STAMPS 131072 768 3072 4 0 3 q4
STAMPS=1 run ${L}_ppstamps 131072 2304 768 1 0 3 pp
STAMPS=1 run ${L}_ppstamps 131072 768 3072 4 0 3 pp
for v in "" _q4NOWAIT _q4NOBAR _q4NODMA _q4NOREAD; do
  STAMPS=1 run $L$v 131072 768 3072 1 0 4 q4
  STAMPS=1 run $L$v 131072 2304 768 1 0 4 q4
done
