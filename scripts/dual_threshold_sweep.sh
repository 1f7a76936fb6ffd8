#!/bin/bash
# where does splitting the T5 stack / the ViT chunks over two streams pay? cold step against batch size with the two thresholds on / off: bash scripts/dual_threshold_sweep.sh
cd "$(dirname "$0")/.."
run() { python bench.py --batch $1 --steps $2 --warmup 6 --no-cpu-baseline --headline-only --live-pmc off ${@:3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for b in 16 32 64 128; do
  st=$(( 640 / b )); [ $st -lt 10 ] && st=10
  for rep in 1 2; do
    echo "batch $b: both dual $(run $b $st)   T5 single $(run $b $st --opt dual_t5_rows=100000000)   ViT single $(run $b $st --opt dual_vit_crops=100000000)   both single $(run $b $st --opt dual_t5_rows=100000000 --opt dual_vit_crops=100000000)   dual_stream=0 $(run $b $st --opt dual_stream=0)"
  done
done
