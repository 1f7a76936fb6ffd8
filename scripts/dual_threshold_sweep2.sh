cd /root/repo
run() { python bench.py --batch $1 --steps $2 --warmup 6 --no-cpu-baseline --headline-only --live-pmc off ${@:3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for b in 20 24 40 48 56; do for rep in 1 2; do
  echo "batch $b: T5 dual $(run $b 20)   T5 single $(run $b 20 --opt dual_t5_rows=100000000)"
done; done
