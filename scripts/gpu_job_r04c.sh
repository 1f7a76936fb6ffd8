# Round 4, job c: cache-policy variants of the ping-pong GEMM (nt on the A stream / on the output stores) and n-group sizes: time + L2-side traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04c}
cd /tmp && export TMPDIR=/tmp
OUT=$O/${TAG}_nt.txt; : > $OUT
run() {  # variant, ngroup_kb, shape args...
  v=$1; kb=$2; shift 2
  L=$R/scripts/micro/gemm_lab_$v
  echo "=== $v ngroup_kb=$kb : $*" >> $OUT
  VIMA_GEMM_NGROUP_KB=$kb timeout 100 $L "$@" 7 pp 2>&1 | grep "median" >> $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pq; VIMA_GEMM_NGROUP_KB=$kb timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pq -o q -- $L "$@" 1 pp > /dev/null 2>&1
    python $R/scripts/pmc_quick.py /tmp/pq $c 2>&1 | grep gemm_pp >> $OUT
  done
}
for v in base nta ntst ntboth; do
  run $v 2560 131072 2304 768 1 0
  run $v 2560 131072 3072 768 1 1
done
run base 1280 131072 2304 768 1 0
run base 1280 131072 3072 768 1 1
run base 8192 131072 2304 768 1 0
run ntboth 1280 131072 2304 768 1 0
run ntboth 8192 131072 2304 768 1 0
run ntboth 8192 131072 3072 768 1 1
run base 2560 131072 768 3072 4 0
run ntboth 2560 131072 768 3072 4 0
cat $OUT
