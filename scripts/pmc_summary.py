"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd SQLite) per kernel and, with a GEMM launch log, per GEMM SHAPE.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads
(guides/MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is used as reported.

The launch log (bench.py --launch-log: [{"kernel", "M", "N", "K"}] of ONE step in launch order, written by the profiled process itself from
vima_prof_read_gemm_launches) attributes per-DISPATCH counter values to shapes: with dual_stream = 0 every step dispatches the same GEMM
sequence, so the GEMM dispatches of the trace are that sequence repeated; a kernel-name mismatch anywhere voids the attribution
(reported, never guessed).

usage: python scripts/pmc_summary.py <fetch.db> <write.db> <out.md> <out.json> [launch_log.json]
       (bench.py imports summarise() for its live collection)"""
import json
import re
import sqlite3
import sys

GEMM_MARKERS = ("gemm_kernel<", "gemm_persistent_kernel", "gemm_pp_kernel", "gemm_wide_kernel", "gemm_q4_kernel", "gemm_resident_kernel")


def short(name):
    s = name.replace("vima::(anonymous namespace)::", "vima::").replace("void ", "")
    return s.split("(")[0][:100]


def _sig(name):
    """what identifies a GEMM kernel in BOTH spellings -- rocprofv3 prints every template argument, the library's launch log abbreviates the tile kernels
    ("vima::gemm_kernel<Tile<128, 128>> act 0"): (base name, tile rows x columns or None, leading template arguments or the activation)"""
    s = short(name)
    base = re.match(r"(?:vima::)?(\w+)", s).group(1)
    t = re.search(r"R?Tile<\s*(\d+),\s*(\d+)", s)
    if t:
        a = re.search(r"act\s*(-?\d+)", s) or re.search(r"Tile<[^>]*>\s*,\s*(-?\d+)", s)
        return base, (int(t.group(1)), int(t.group(2))), (a.group(1),) if a else ()
    args = re.search(r"<([^>]*)>", s)
    return base, None, tuple(x.strip() for x in args.group(1).split(",")) if args else ()


def _same_kernel(a, b):
    (ba, ta, aa), (bb, tb, ab) = _sig(a), _sig(b)
    n = min(len(aa), len(ab))
    return ba == bb and (ta == tb or ta is None or tb is None) and aa[:n] == ab[:n]


def per_dispatch(db, counter):
    """[(order key, kernel_name, value)] one entry per dispatch, in dispatch order (value = mean of the dispatch's rows, which is what the
    per-kernel averages of the earlier summaries used)"""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    key = next((k for k in ("dispatch_id", "id", "start", "start_timestamp") if k in cols), None)
    if key is None:
        rows = list(c.execute("select rowid, kernel_name, value from counters_collection where counter_name=? order by rowid", (counter,)))
        return rows, cols
    rows = list(c.execute(f"select {key}, kernel_name, avg(value) from counters_collection where counter_name=? group by {key}, kernel_name "
                          f"order by {key}", (counter,)))
    return rows, cols


def summarise(fetch_db, write_db, launch_log=None):
    fd, cols = per_dispatch(fetch_db, "FETCH_SIZE")
    wd, _ = per_dispatch(write_db, "WRITE_SIZE")
    agg = {}
    for rows, idx in ((fd, 0), (wd, 1)):
        for _, k, v in rows:
            a = agg.setdefault(k, [[0, 0.0], [0, 0.0]])
            a[idx][0] += 1
            a[idx][1] += v
    table = []
    for k, ((nf, tf), (nw, tw)) in agg.items():
        af, aw = (tf / nf if nf else 0.0), (tw / nw if nw else 0.0)
        table.append((short(k), nf or nw, af, aw, (2 * af + aw) * 1024))
    table.sort(key=lambda r: -r[1] * r[4])
    gem = [r for r in table if any(m in r[0] for m in GEMM_MARKERS) and "gemm_resident" not in r[0]]
    n = sum(r[1] for r in gem)
    gem_bytes = sum(r[1] * r[4] for r in gem) / max(n, 1)
    out = {"gemm_bf16_launches": n, "gemm_bf16_bytes_per_launch": gem_bytes,
           "per_kernel": {r[0]: {"launches": r[1], "bytes_per_launch": r[4]} for r in table[:40]},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0",
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-count)", "counters_collection_columns": cols,
           "_table": table}
    if launch_log:
        try:
            log = json.load(open(launch_log))
            out["per_shape"], out["per_shape_note"] = per_shape(fd, wd, log)
        except Exception as e:   # noqa: BLE001
            out["per_shape"], out["per_shape_note"] = None, f"launch log not usable: {type(e).__name__}: {e}"
    return out


def per_shape(fd, wd, log):
    """join per-dispatch FETCH / WRITE values with the per-step launch log by position among the GEMM dispatches"""
    def gemm_seq(rows):
        return [(k, v) for _, k, v in rows if any(m in k for m in GEMM_MARKERS)]
    fs, ws = gemm_seq(fd), gemm_seq(wd)
    L = len(log)
    if L == 0 or len(fs) == 0 or len(fs) % L or len(ws) != len(fs):
        return None, (f"{len(fs)} / {len(ws)} GEMM dispatches in the FETCH / WRITE passes are not a multiple of the {L} launches of one "
                      "logged step: no per-shape attribution")
    acc = {}
    for i, ((kf, vf), (kw, vw)) in enumerate(zip(fs, ws)):
        e = log[i % L]
        if not _same_kernel(e["kernel"], kf) or short(kf) != short(kw):
            return None, f"dispatch {i}: trace kernel {short(kf)} vs logged {e['kernel']} (write pass: {short(kw)}): sequence mismatch"
        key = (short(kf), e["M"], e["N"], e["K"])
        a = acc.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += vf
        a[2] += vw
    rows = []
    for (k, M, N, K), (cnt, tf, tw) in acc.items():
        rows.append({"kernel": k, "M": M, "N": N, "K": K, "launches_per_step": cnt * L // len(fs) if len(fs) >= L else cnt,
                     "fetch_mb": 2 * tf / cnt * 1024 / 1e6, "write_mb": tw / cnt * 1024 / 1e6, "bytes_per_launch": (2 * tf + tw) / cnt * 1024})
    rows.sort(key=lambda r: -r["bytes_per_launch"] * r["launches_per_step"])
    return rows, f"{len(fs) // L} step(s) x {L} GEMM launches matched by dispatch order"


def algorithmic_bytes(kernel, M, N, K):
    """bf16 operands once + output (+ bf16 residual read for the stream epilogue EPI 4; + gate read for EPI 2)"""
    b = 2.0 * (M * K + N * K + M * N)
    if ", 4" in kernel:
        b += 2.0 * M * N + 4.0 * M * (N // 32)
    if ", 2" in kernel and "pp_kernel" in kernel or ", 2>" in kernel:
        b += 2.0 * M * N
    if ", 3" in kernel and ("pp_kernel" in kernel or "persistent" in kernel):
        b = 2.0 * (M * K + N * K) + 4.0 * M * N * 2 + 2.0 * M * N
    return b


def write_md(summ, out_md, title=None):
    with open(out_md, "w") as fh:
        fh.write((title or "# HBM traffic per kernel launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs)") + "\n\n")
        fh.write("bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE half-count correction per MI355X_MICROARCH.md)\n\n")
        fh.write("| kernel | launches | avg FETCH_SIZE KiB | avg WRITE_SIZE KiB | corrected MB / launch |\n|---|---|---|---|---|\n")
        for r in summ["_table"][:25]:
            fh.write(f"| `{r[0]}` | {r[1]} | {r[2]:.0f} | {r[3]:.0f} | {r[4] / 1e6:.1f} |\n")
        fh.write(f"\nAll bf16 GEMM instantiations (`gemm_kernel<bf16,...>`, `gemm_pp_kernel`, `gemm_persistent_kernel`, `gemm_wide_kernel`, `gemm_q4_kernel`): "
                 f"{summ['gemm_bf16_launches']} launches, {summ['gemm_bf16_bytes_per_launch'] / 1e6:.1f} MB per launch on average.\n")
        if summ.get("per_shape"):
            fh.write(f"\n## Per GEMM shape ({summ['per_shape_note']})\n\nalgorithmic = each bf16 operand once + output (+ residual / gate read, RMS partials)\n\n")
            fh.write("| kernel | M | N | K | launches / step | fetch MB | write MB | counted MB | algorithmic MB | counted / algorithmic |\n|---|---|---|---|---|---|---|---|---|---|\n")
            for r in summ["per_shape"][:30]:
                alg = algorithmic_bytes(r["kernel"], r["M"], r["N"], r["K"])
                fh.write(f"| `{r['kernel']}` | {r['M']} | {r['N']} | {r['K']} | {r['launches_per_step']} | {r['fetch_mb']:.1f} | {r['write_mb']:.1f} | "
                         f"{r['bytes_per_launch'] / 1e6:.1f} | {alg / 1e6:.1f} | {r['bytes_per_launch'] / alg:.2f} |\n")
        elif "per_shape_note" in summ:
            fh.write(f"\nPer-shape attribution: {summ['per_shape_note']}\n")
        fh.write(f"\n(`counters_collection` columns of this rocprofv3: {', '.join(summ['counters_collection_columns'])})\n")


def main():
    fdb, wdb, out_md, out_json = sys.argv[1:5]
    summ = summarise(fdb, wdb, sys.argv[5] if len(sys.argv) > 5 else None)
    write_md(summ, out_md)
    j = {k: v for k, v in summ.items() if k != "_table"}
    json.dump(j, open(out_json, "w"), indent=1)
    print("gemm bf16:", summ["gemm_bf16_launches"], "launches,", summ["gemm_bf16_bytes_per_launch"] / 1e6, "MB/launch;", summ.get("per_shape_note", ""))


if __name__ == "__main__":
    main()
