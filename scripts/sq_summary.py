"""Summarise an SQ-counter rocprofv3 pass per kernel: MFMA utilisation, wait fractions, LDS conflicts.
usage: python scripts/sq_summary.py <db> <out.md>"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    data = {}
    for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        data.setdefault(k, {})[cn] = (n, avg)
    rows = []
    for k, v in data.items():
        g = lambda name: v.get(name, (0, 0.0))[1]
        n = v.get("SQ_WAVE_CYCLES", (0, 0))[0]
        gui = g("GRBM_GUI_ACTIVE") / 8.0            # summed over the 8 XCDs
        simd_cycles = gui * 1024.0                   # 256 CUs x 4 SIMDs
        rows.append((k, n, gui, g("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles if simd_cycles else 0,
                     g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                     g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_LDS_IDX_ACTIVE") / (gui * 256.0) if gui else 0))
    rows.sort(key=lambda r: -r[1] * r[2])
    with open(out, "w") as f:
        f.write("# SQ counters per kernel (rocprofv3 --pmc, one pass)\n\n")
        f.write("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); waits as fraction of SQ_WAVE_CYCLES; "
                "LDS busy = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)\n\n")
        f.write("| kernel | launches | avg cycles | MfmaUtil | WAIT_ANY | WAIT_INST_ANY | LDS conflict / active | LDS busy |\n|---|---|---|---|---|---|---|---|\n")
        for k, n, gui, mf, wa, wi, lc, lb in rows[:16]:
            short = k.replace("vima::(anonymous namespace)::", "vima::").replace("void ", "").split("(")[0][:100]
            f.write(f"| `{short}` | {n} | {gui:.0f} | {mf:.3f} | {wa:.3f} | {wi:.3f} | {lc:.3f} | {lb:.3f} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
