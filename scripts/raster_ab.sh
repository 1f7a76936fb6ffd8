#!/bin/bash
# Raster A/B of the K = 768 wide-N GEMMs: shipped n-group raster vs ONE group (A fetched once) vs half-size groups, each with
# its launch time and its FETCH_SIZE / WRITE_SIZE per launch (two rocprofv3 --pmc passes).  bash scripts/raster_ab.sh > gpurun_out/<tag>_raster_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
for shape in "131072 3072 768 1" "131072 2304 768 0" "81920 3072 768 3"; do
  for kb in 2560 1000000 1300; do
    echo "=== shape $shape ngroup_kb=$kb"
    for rep in 1 2; do VIMA_GEMM_NGROUP_KB=$kb python $R/scripts/raster_ab.py $shape 2>/dev/null | grep "^M"; done
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/rab_$c
      (cd /tmp && VIMA_GEMM_NGROUP_KB=$kb timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/rab_$c -o t -- python $R/scripts/raster_ab.py $shape 7 > /dev/null 2>&1)
      python $R/scripts/pmc_quick.py /tmp/rab_$c $c | grep gemm_pp
    done
  done
done
