# L2 hit rate of the ping-pong GEMM at the model's shapes (lab binary, no python)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for shape in "65536 2304 768 1 0" "65536 768 3072 4 0" "65536 3072 768 1 1"; do
  rm -rf /tmp/l2prof
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/l2prof -o l2 -- $R/scripts/micro/gemm_lab $shape 3 pp > /tmp/l2_stdout.txt 2>&1
  echo "== $shape"; grep "median" /tmp/l2_stdout.txt | tail -1
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/l2prof/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r.get('Kernel_Name', '')
        if 'gemm_pp' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    hit, miss = d.get('TCC_HIT_sum', 0), d.get('TCC_MISS_sum', 0)
    per = {c: v / max(n[(k, c)], 1) for c, v in d.items()}
    print(k[:60], {c: f'{v:.3e}' for c, v in per.items()}, 'hit rate %.3f' % (hit / max(hit + miss, 1)))
PY
done
