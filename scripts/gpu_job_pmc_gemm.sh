R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in 2 1; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_gemm_t$T -o g -- python $R/scripts/gemm_micro.py 131072 2304 768 $T 3 > $R/gpurun_out/pmc_gemm_t$T.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gemm2_t$T -o g -- python $R/scripts/gemm_micro.py 131072 2304 768 $T 3 > $R/gpurun_out/pmc_gemm2_t$T.txt 2>&1
done
ls $R/gpurun_out/pmc_gemm_t2 $R/gpurun_out/pmc_gemm2_t2
