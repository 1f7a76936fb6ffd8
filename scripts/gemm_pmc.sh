#!/bin/bash
# busy / stall counters of the vector-memory path (TA -> TCP -> TCC) for the 256x256 ping-pong GEMM and the 256x384 four-wave GEMM,
# lab binary, uniform random operands. Separate small --pmc passes (no trace domains besides --kernel-trace), each under its own time limit: on this
# pool some counter sets do not return. Usage: bash scripts/gemm_pmc.sh [M N K epi act]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/scripts/micro/gemm_lab
SHAPE=${*:-32768 3072 768 1 1}
export TMPDIR=/tmp; cd /tmp
pass() {  # pass <name> <counters...>
  local name=$1; shift
  rm -rf /tmp/gp_$name
  INNER=1 timeout -k 5 75 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/gp_$name -o g --output-format csv -- $L $SHAPE 1 pp,q4 > /tmp/gp_$name.log 2>&1
  local rc=$?
  python3 - "$name" "$rc" "$*" <<'PY'
import csv, glob, sys, collections
name, rc, ctrs = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(f"/tmp/gp_{name}/**/*counter_collection.csv", recursive=True)
if not f: print(f"  pass {name} ({ctrs}): no counter file (rocprofv3 exit code {rc})"); sys.exit(0)
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
echo "# gemm_lab $SHAPE, per-dispatch means (counter values are summed over the chip by rocprofv3)"
pass grbm GRBM_GUI_ACTIVE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass tcc1 TCC_HIT TCC_MISS TCC_REQ
pass tcc2 TCC_BUSY TCC_TAG_STALL
pass tcc3 TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ TCC_EA0_WRREQ
pass tcp1 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ
pass tcp2 TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_GATE_EN1
pass ta1 TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pass ta2 TA_FLAT_READ_LDS_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES
