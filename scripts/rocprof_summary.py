"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd SQLite output) into a small markdown/CSV table that can be
committed under profiles/.  usage: python scripts/rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    extra = {}
    for name, vg, ag, sg, lds, wg, gx in c.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
            "from kernels group by name"):
        extra[name] = (vg, ag, sg, lds, wg, gx)
    with open(out, "w") as f:
        f.write(f"# {title}\n\n")
        f.write("Source: `rocprofv3 --kernel-trace --stats` (rocpd database `top_kernels` view); durations in microseconds.\n\n")
        f.write(f"Total kernel time: {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches.\n\n")
        f.write("| kernel | calls | total us | avg us | % | vgpr | agpr | sgpr | lds B | wg |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("vima::(anonymous namespace)::", "vima::").replace("void ", "")
            short = short.split("(")[0][:110]
            e = extra.get(name, ("", "", "", "", "", ""))
            f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} | {e[0]} | {e[1]} | {e[2]} | {e[3]} | {e[4]} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
