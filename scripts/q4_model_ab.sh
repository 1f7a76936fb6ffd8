#!/bin/bash
# wall-clock A/B of the gemm_q4 modes in the model (two streams, as benchmarked), alternating on one box: bash scripts/q4_model_ab.sh "0 1 4 5" [reps] [bench args]
cd "$(dirname "$0")/.."
MODES=${1:-"0 1 4 5"}; REPS=${2:-3}; shift 2
for rep in $(seq $REPS); do for m in $MODES; do
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --headline-only --live-pmc off --opt gemm_q4=$m "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_q4=$m', d['ms_per_step'])"
done; done
