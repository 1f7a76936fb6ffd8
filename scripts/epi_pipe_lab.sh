#!/bin/bash
# bf16-output epilogue staged through LDS in bf16 (round 6): old (fp32 slab) vs new lab binary, alternating, stamps; then bit identity of the new pp kernel against
# gemm_persistent_kernel / the one-tile kernel, which keep the fp32-slab epilogue. old = the commit before, new = this tree (scripts/micro/build_gemm_lab.sh each).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
for shape in "131072 2304 768 1 0" "131072 3072 768 1 1" "81920 3072 768 1 3" "131072 1536 768 1 0"; do
  for v in old new; do
    echo -n "$v: "; STAMPS=1 timeout 200 $R/scripts/micro/gemm_lab_$v $shape 5 pp 2>&1 | grep "TFLOP\|stamps" | sed 's/.*main loop \([0-9]*\) (\([0-9]*\) per K-tile)  epilogue \([0-9]*\).*/main \1 epi \3 |/' | tr '\n' ' '; echo
  done
done
done
echo "== bit identity (new pp vs persist / tile; bias + QuickGELU, ReLU, plain)"
BIAS=1 $R/scripts/micro/gemm_lab_new 16384 3072 768 1 3 2 pp,persist,tile | grep "differ\|fp64"
$R/scripts/micro/gemm_lab_new 16384 3072 768 1 1 2 pp,persist,tile | grep "differ\|fp64"
$R/scripts/micro/gemm_lab_new 16384 2304 768 1 0 2 pp,persist,tile | grep "differ\|fp64"
BIAS=1 $R/scripts/micro/gemm_lab_new 81920 2304 768 1 0 2 pp,persist | grep "differ\|fp64"
