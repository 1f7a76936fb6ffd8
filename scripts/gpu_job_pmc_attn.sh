# raw SQ counters of the T5 attention kernel (two PMC passes), summarised on the box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/scripts/attn_micro.py 256 12 512 64 3 | tail -1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pa1 -o g -- python $R/scripts/attn_micro.py 256 12 512 64 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d /tmp/pa2 -o g -- python $R/scripts/attn_micro.py 256 12 512 64 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_FLAT -d /tmp/pa3 -o g -- python $R/scripts/attn_micro.py 256 12 512 64 2 > /dev/null 2>&1
python - > $R/gpurun_out/attn_pmc.txt <<'PY'
import sqlite3, glob
for d in ("/tmp/pa1", "/tmp/pa2", "/tmp/pa3"):
    for db in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%attn_mfma4%' group by kernel_name, counter_name"):
            print(f"{cn:28s} n={n} avg={avg:.0f}")
PY
cat $R/gpurun_out/attn_pmc.txt
