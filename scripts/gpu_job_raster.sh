R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for RA in 0 1 2; do
export RASTER=$RA
for SH in "131072 2304 768" "131072 768 3072" "81920 3072 768"; do
python $R/scripts/gemm_micro.py $SH 2 5 2>&1 | tail -1
done
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/tcc_r$RA -o g -- python $R/scripts/gemm_micro.py 131072 2304 768 2 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/fetch_r$RA -o g -- python $R/scripts/gemm_micro.py 131072 2304 768 2 2 > /dev/null 2>&1
done
