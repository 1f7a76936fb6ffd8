# Round 4, job a: (1) epilogue / tile-boundary ablations of the ping-pong GEMM (lab binaries, timing only), (2) the small-M A/B queued at the
# end of round 3 (64x64 resident tile with two chunk buffers = two workgroups per CU), (3) clock state of microsecond kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04a}
cd $R
export STAMPS=1
for v in base nores nostore skew3; do
  L=scripts/micro/gemm_lab_$v
  echo "=== $v" >> $O/${TAG}_lab.txt
  timeout 100 $L 131072 768 768 4 0 5 pp >> $O/${TAG}_lab.txt 2>&1
  timeout 100 $L 131072 768 3072 4 0 5 pp >> $O/${TAG}_lab.txt 2>&1
  timeout 100 $L 131072 2304 768 1 0 5 pp 2>&1 | grep -v "host fp64" >> $O/${TAG}_lab.txt
  timeout 100 $L 81920 768 768 4 0 5 pp >> $O/${TAG}_lab.txt 2>&1
done
grep -v "clocks per" $O/${TAG}_lab.txt
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30 > $O/${TAG}_smi.txt
timeout 600 python scripts/small_m_ab.py micro 20 > $O/${TAG}_small_m_ab.txt 2>&1; tail -45 $O/${TAG}_small_m_ab.txt
# small kernels under a pinned performance level (experiment: documents what the DPM state costs a batch-1 step)
timeout 200 python bench.py --batch 1 --steps 50 --warmup 10 --headline-only --no-cpu-baseline > $O/${TAG}_b1_auto.json 2> $O/${TAG}_b1_auto.err
rocm-smi --setperflevel high >> $O/${TAG}_smi.txt 2>&1
timeout 200 python bench.py --batch 1 --steps 50 --warmup 10 --headline-only --no-cpu-baseline > $O/${TAG}_b1_high.json 2> $O/${TAG}_b1_high.err
timeout 200 python bench.py --batch 1 --steps 50 --warmup 10 --headline-only --no-cpu-baseline --opt graphs=1 > $O/${TAG}_b1_high_graphs.json 2> $O/${TAG}_b1_high_graphs.err
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30 >> $O/${TAG}_smi.txt
rocm-smi --setperflevel auto >> $O/${TAG}_smi.txt 2>&1
for f in b1_auto b1_high b1_high_graphs; do python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], "ms")
except Exception as e: print("$f failed", e)
PY
done
cat $O/${TAG}_smi.txt | tail -40
