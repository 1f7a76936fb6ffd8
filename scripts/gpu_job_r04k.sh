R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04k}
cd $R; export STAMPS=1
for rep in 1 2 3; do
  for cfg in "base 1" "wexact 1"; do set -- $cfg
  if [ $2 = 1 ]; then export SSQ=1; else unset SSQ; fi
  L=scripts/micro/gemm_lab_$1
  echo "=== $1 SSQ=$2" >> $O/${TAG}_lab.txt
  timeout 100 $L 131072 768 768 4 0 7 persist,pp >> $O/${TAG}_lab.txt 2>&1
  timeout 100 $L 131072 768 3072 4 0 7 persist,pp >> $O/${TAG}_lab.txt 2>&1
done; done
unset SSQ
echo "=== base SSQ=0 (other epilogues, pinned stores)" >> $O/${TAG}_lab.txt
timeout 100 scripts/micro/gemm_lab_base 131072 2304 768 1 0 5 persist,pp 2>&1 | grep -v "host fp64" >> $O/${TAG}_lab.txt
timeout 100 scripts/micro/gemm_lab_base 131072 768 768 4 0 5 persist,pp >> $O/${TAG}_lab.txt 2>&1
timeout 100 scripts/micro/gemm_lab_base 65536 768 768 3 0 5 persist,pp >> $O/${TAG}_lab.txt 2>&1
grep -v "clocks per\|stamps" $O/${TAG}_lab.txt
