#!/bin/bash
# SQ counters of the T5 attention kernel alone (scripts/attn_micro.py, T5 shape), one small rocprofv3 --pmc pass per group -> gpurun_out/attn_pmc.txt
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
LIB=${1:-vima_amd/lib/libvima_hip.so}
OUT=$O/attn_pmc.txt; : > $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counter_names.txt)
pass() {
  rm -rf /tmp/apmc
  (cd /tmp && VIMA_HIP_LIB=$R/$LIB timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/apmc -o a -- python $R/scripts/attn_micro.py 256 12 512 64 3 > /dev/null 2>&1)
  python $R/scripts/pmc_quick.py /tmp/apmc "$@" 2>&1 | grep "attn_mfma4\|no \*_results" >> $OUT
}
pass SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES
pass SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS
pass SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16
cat $OUT
