R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/scripts/attn_micro.py 256 12 512 64 3 | tail -1
MODE=1 LQ=8 python $R/scripts/attn_micro.py 256 24 512 32 3 | tail -1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_attn1 -o g -- python $R/scripts/attn_micro.py 256 12 512 64 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_attn2 -o g -- python $R/scripts/attn_micro.py 256 12 512 64 2 > /dev/null 2>&1
