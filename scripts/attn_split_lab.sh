#!/bin/bash
# split-key cross attention (decoder, 8 queries x 512 prompt keys, head-major K / V as in the model is not reachable from the op API: row-major here), two libraries alternating
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for t in ${1:-base sets4}; do echo "== $t"; VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so MODE=1 LQ=8 MASKF=0.1 CHECK=1 timeout 120 python scripts/attn_micro.py 256 24 512 32 30 2>&1 | tail -2; done; done
for t in ${1:-base sets4}; do
VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so MODE=1 LQ=8 CHECK=1 timeout 120 python scripts/attn_micro.py 16 24 1500 32 3 2>&1 | tail -2
VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so MODE=1 LQ=8 CHECK=1 timeout 120 python scripts/attn_micro.py 16 24 300 32 3 2>&1 | tail -2
done
