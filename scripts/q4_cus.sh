#!/bin/bash
# (1) Is a tile's epilogue / first-K-tile cost a per-CU or a chip-wide limit?  The same GEMM on 256 / 128 / 64 CUs (lab knob VIMA_GEMM_LAB_CUS), stamps per tile.
# (2) The counters available for the L2 <-> CU path on this box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
run() { echo "== $*"; timeout 200 "$@" 2>&1 | grep -v "^$\|host fp64"; }
for cus in 256 128 64 32; do
  export VIMA_GEMM_LAB_CUS=$cus
  echo "#### CUs = $cus"
  STAMPS=1 run $L 32768 2304 768 1 0 3 pp,q4
  STAMPS=1 run $L 32768 768 3072 4 0 3 pp,q4
done
unset VIMA_GEMM_LAB_CUS
echo "#### counters"
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TA\|TCC\|TD\|SQ_INSTS_VALU_MFMA\|SQ_BUSY\|GRBM\)_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | fold -w 220
