R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04l}
cd $R; export STAMPS=1
OUT=$O/${TAG}_lab.txt; : > $OUT
run() { echo "=== $1 SSQ=${SSQ:-0} : $2 $3 $4 epi$5 act$6" >> $OUT; timeout 120 scripts/micro/gemm_lab_$1 $2 $3 $4 $5 $6 7 $7 >> $OUT 2>&1; }
for rep in 1 2; do
  for v in old base; do
    export SSQ=1; run $v 131072 768 768 4 0 tile,persist,pp; run $v 131072 768 3072 4 0 tile,pp
    unset SSQ;    run $v 131072 768 768 4 0 tile,pp; run $v 81920 768 768 4 0 tile,pp
    run $v 65536 768 768 3 0 tile,persist,pp; run $v 65536 768 3072 3 0 tile,pp
    run $v 65536 3072 768 2 2 tile,persist,pp
    run $v 131072 2304 768 1 0 tile,pp
  done
done
grep -v "clocks per\|host fp64" $OUT | grep -v "stamps"
