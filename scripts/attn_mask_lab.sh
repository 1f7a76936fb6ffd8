#!/bin/bash
# T5 attention with 10 % scattered masked keys (the benchmark's prompts), two libraries alternating: bash scripts/attn_mask_lab.sh "base cinit"
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for t in ${1:-base cinit}; do echo "== $t"
  MASKF=0.1 CHECK=1 VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so timeout 120 python scripts/attn_micro.py 256 12 512 64 20 2>&1 | tail -2
  MASKF=0.1 CHECK=1 VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so timeout 120 python scripts/attn_micro.py 64 12 1024 64 20 2>&1 | tail -2
done; done
for t in ${1:-base cinit}; do echo "== $t cross, D 32 / ragged T5"
  MASKF=0.1 CHECK=1 MODE=1 LQ=128 VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so timeout 120 python scripts/attn_micro.py 32 24 512 32 5 2>&1 | tail -2
  MASKF=0.3 CHECK=1 VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so timeout 120 python scripts/attn_micro.py 8 12 300 64 5 2>&1 | tail -2
done
