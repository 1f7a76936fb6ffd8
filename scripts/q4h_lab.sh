#!/bin/bash
# gemm_q4_kernel<.., MIH = 1> (128x384 tile, "q4h") vs gemm_pp_kernel (256x256) on the batch-32 shapes whose 256-row tilings leave CUs idle; bit identity on small and large problems
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
N=${1:-7}
run() { echo "== gemm_lab $*"; timeout 300 $L "$@" 2>&1 | grep -v "^$\|clocks per"; }
run 8192 3072 768 1 0 3 pp,q4h
BIAS=1 run 16384 768 768 4 0 3 pp,q4h
STAMPS=1 run 16384 768 768 4 0 $N pp,q4h,q4      # T5 o at batch 32: 192 tiles of 256x256 / 256 of 128x384 / 128 of 256x384
STAMPS=1 run 16384 768 3072 4 0 $N pp,q4h        # T5 wo at batch 32
STAMPS=1 run 16384 2304 768 1 0 $N pp,q4h,q4     # T5 q|k|v at batch 32 (576 / 768 / 384 tiles)
STAMPS=1 run 16384 3072 768 1 1 $N pp,q4h,q4     # T5 wi
BIAS=1 run 10240 768 3072 4 0 $N pp,q4h          # ViT c_proj at batch 32 (M = 2048 crops x 5)
BIAS=1 run 10240 3072 768 1 3 $N pp,q4h          # ViT fc
STAMPS=1 run 131072 2304 768 1 0 5 pp,q4h,q4     # headline shape for reference
