R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04i}
cd $R; export STAMPS=1 SSQ=1
for rep in 1 2; do for v in base wssq; do
  L=scripts/micro/gemm_lab_$v
  echo "=== $v (SSQ=1)" >> $O/${TAG}_lab.txt
  timeout 100 $L 131072 768 768 4 0 7 persist,pp >> $O/${TAG}_lab.txt 2>&1
  timeout 100 $L 131072 768 3072 4 0 7 pp >> $O/${TAG}_lab.txt 2>&1
done; done
grep -v "clocks per" $O/${TAG}_lab.txt
