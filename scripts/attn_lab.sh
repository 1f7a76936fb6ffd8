#!/bin/bash
# Round 6 attention lab: variants of attn_mfma4_kernel built by scripts/build_attn_variants.sh (-DVIMA_ATTN_X=<bits>), alternating on one box.
#   usage (on the GPU box): bash scripts/attn_lab.sh "<tags>" [reps]        -> gpurun_out/attn_lab.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/attn_lab.txt
: > $OUT
TAGS=${1:-"base p1 p12"}
REPS=${2:-2}
for rep in $(seq $REPS); do
  for t in $TAGS; do
    echo "== $t" >> $OUT
    chk=1; case $t in nodma|notab) chk="";; esac
    qg=1; lib=$t; case $t in *:qg2) qg=2; lib=${t%%:*};; esac   # tag:qg2 = 64 queries per wave (option attn_qg = 2)
    QG=$qg VIMA_HIP_LIB=build_ablate/libvima_hip_$lib.so CHECK=$chk timeout 120 python scripts/attn_micro.py 256 12 512 64 20 2>&1 | grep -v "^$" | tail -2 >> $OUT
    QG=$qg VIMA_HIP_LIB=build_ablate/libvima_hip_$lib.so CHECK=$chk timeout 120 python scripts/attn_micro.py 64 12 1024 64 20 2>&1 | tail -2 >> $OUT
  done
done
for t in stamps k3stamps; do
  [ -f build_ablate/libvima_hip_$t.so ] || continue
  echo "== $t" >> $OUT
  VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so STAMPS=1 timeout 120 python scripts/attn_micro.py 256 12 512 64 5 2>&1 | tail -2 >> $OUT
  VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so STAMPS=1 timeout 120 python scripts/attn_micro.py 64 12 1024 64 5 2>&1 | tail -2 >> $OUT
done
cat $OUT
