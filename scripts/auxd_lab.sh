#!/bin/bash
# variant binaries: for d in 1 2 3 4; do bash scripts/micro/build_gemm_lab.sh r06 auxd$d -DVIMA_LAB_AUXD=$d; done
# stream epilogue (EPI 4) / gate epilogue (EPI 2): prefetch distance of the per-row operand, 1 (rounds 2-5) .. 4 slabs; separate binaries, rounds interleaved by the loop below
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
for rep in 1 2; do
for shape in "131072 768 768 4 0" "131072 768 3072 4 0" "81920 768 768 4 0" "131072 3072 768 2 3"; do
  for d in 1 2 3 4; do
    echo -n "auxd=$d: "; STAMPS=1 timeout 200 ${L}_auxd$d $shape 5 pp 2>&1 | grep "TFLOP\|stamps" | sed 's/.*epilogue \([0-9]*\).*/epilogue \1 clocks |/' | tr '\n' ' '; echo
  done
done
done
