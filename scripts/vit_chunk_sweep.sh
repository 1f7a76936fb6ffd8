#!/bin/bash
# headline ms per step against the ViT chunk size (option vit_chunk; crops per chunk), same box: bash scripts/vit_chunk_sweep.sh
cd "$(dirname "$0")/.."
for rep in 1 2; do for c in 16384 8192 10923 13654 20480 27307 40960; do
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline --headline-only --live-pmc off --opt vit_chunk=$c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vit_chunk $c', d['ms_per_step'])"
done; done
