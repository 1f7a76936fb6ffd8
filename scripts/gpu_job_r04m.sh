R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04m}
cd $R
OUT=$O/${TAG}_lab.txt; : > $OUT
for vs in tile persist pp; do echo "== $vs small" >> $OUT; timeout 60 scripts/micro/gemm_lab_base 2048 768 768 4 0 1 $vs >> $OUT 2>&1; echo "rc $?" >> $OUT; done
for vs in persist pp; do echo "== $vs T5o" >> $OUT; timeout 60 scripts/micro/gemm_lab_base 131072 768 768 4 0 2 tile,$vs >> $OUT 2>&1; echo "rc $?" >> $OUT; done
echo "== epi3" >> $OUT; timeout 60 scripts/micro/gemm_lab_base 65536 768 768 3 0 3 tile,persist,pp >> $OUT 2>&1
echo "== epi2" >> $OUT; timeout 60 scripts/micro/gemm_lab_base 65536 3072 768 2 2 3 tile,persist,pp >> $OUT 2>&1
cat $OUT
