#!/bin/bash
# busy / stall counters of the vector-memory path (TA -> TCP -> TCC) for the 256x256 ping-pong GEMM and the 256x384 four-wave GEMM on the
# T5 wi shape (131072 x 3072 x 768, ReLU, bf16 out), lab binary, uniform random operands. Separate --pmc passes (no trace domains besides --kernel-trace).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
export TMPDIR=/tmp; cd /tmp
pass() {  # pass <name> <counters...>
  local name=$1; shift
  rm -rf /tmp/gp_$name
  INNER=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/gp_$name -o g --output-format csv -- $L 131072 3072 768 1 1 2 pp,q4 > /tmp/gp_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
f = glob.glob(f"/tmp/gp_{name}/**/*counter_collection.csv", recursive=True)
if not f: print(name, "no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "gemm_pp_kernel" in k: k = "pp"
    elif "gemm_q4_kernel" in k: k = "q4"
    else: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"  [{k}] " + "  ".join(f"{c} {sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())) + f"   ({len(next(iter(acc[k].values())))} dispatches)")
PY
}
echo "# per-dispatch means over the timed launches (counter values are summed over the chip by rocprofv3)"
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
pass ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES TA_FLAT_READ_LDS_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS
pass tcp TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1 TCP_GATE_EN2
pass tcc1 TCC_BUSY TCC_CYCLE TCC_TAG_STALL TCC_REQ
pass tcc2 TCC_HIT TCC_MISS TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL
pass tcc3 TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_SRC_FIFO_FULL
