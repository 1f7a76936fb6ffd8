R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04j}
cd $R; export STAMPS=1
for rep in 1 2; do
  for cfg in "base 0" "base 1" "sqpair 1"; do set -- $cfg
  if [ $2 = 1 ]; then export SSQ=1; else unset SSQ; fi
  L=scripts/micro/gemm_lab_$1
  echo "=== $1 SSQ=$2" >> $O/${TAG}_lab.txt
  timeout 100 $L 131072 768 768 4 0 7 persist,pp >> $O/${TAG}_lab.txt 2>&1
  timeout 100 $L 131072 768 3072 4 0 7 pp >> $O/${TAG}_lab.txt 2>&1
done; done
grep -v "clocks per" $O/${TAG}_lab.txt
