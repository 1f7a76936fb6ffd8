#!/bin/bash
# Lab A/B of gemm_q4_kernel (256x384, four waves) against gemm_pp_kernel (256x256, eight waves) on the headline GEMM shapes: interleaved rounds in one
# process per shape, bitwise comparison, per-tile stamps (scripts/micro/gemm_lab). Usage: bash scripts/q4_lab.sh [rounds] [variants]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
N=${1:-7}; V=${2:-pp,q4}
run() { echo "== gemm_lab $*"; timeout 300 $L "$@" 2>&1 | grep -v "^$"; }
# small shapes first: correctness (host fp64 check + bit identity), then the headline shapes
run 4096 3072 768 1 0 3 $V
STAMPS=1 run 131072 3072 768 1 1 $N $V     # T5 wi (ReLU)
STAMPS=1 run 131072 2304 768 1 0 $N $V     # T5 q|k|v
STAMPS=1 run 131072 768 3072 4 0 $N $V     # T5 wo (bf16 stream)
STAMPS=1 run 131072 768 768 4 0 $N $V      # T5 o (bf16 stream)
BIAS=1 run 81920 3072 768 1 3 $N $V        # ViT fc (QuickGELU, bias)
BIAS=1 run 81920 2304 768 1 0 $N $V        # ViT in_proj (bias)
BIAS=1 run 81920 768 3072 4 0 $N $V        # ViT c_proj (bf16 stream, bias)
SSQ=1 run 131072 768 3072 4 0 $N $V        # T5 wo with RMS partials
run 8192 8192 8192 1 0 $N $V
