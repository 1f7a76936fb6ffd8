"""GEMM microbenchmark through the C ABI (vima_op_linear): python scripts/gemm_micro.py M N K [tile] [iters]
Used under rocprofv3 --pmc to read MFMA / LDS counters for the GEMM kernel alone."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402


def main():
    M, N, K = (int(x) for x in sys.argv[1:4])
    tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
    pol._ensure_handle()
    pol.set_option("gemm_tile", tile)
    pol.set_option("gemm_wide", int(os.environ.get("WIDE", "0")))       # 1: 256x384 persistent tile
    pol.set_option("gemm_raster", int(os.environ.get("RASTER", "0")))
    pol.set_option("gemm_epi", int(os.environ.get("EPI", "1")))
    pol.set_option("op_bf16_out", int(os.environ.get("BF16OUT", "0")))   # 1: bf16-only output like most in-model GEMMs
    pol.set_option("op_stream_T", int(os.environ.get("STREAMT", "0")))   # 1 (with RES=1): residual + output in bf16 (EPI 4)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    out = torch.empty(M, N, device="cuda")
    res = torch.randn(M, N, device="cuda") if os.environ.get("RES") else None       # residual epilogue (fp32 in / fp32 out)
    bias = torch.randn(N, device="cuda") if os.environ.get("BIAS") else None
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    for _ in range(iters):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), p(bias), None, p(res), M, N, K, int(os.environ.get('ACT', '0')), p(out), pol._stream()))
    torch.cuda.synchronize()
    pol.prof_enable(True)
    for _ in range(iters):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), p(bias), None, p(res), M, N, K, int(os.environ.get('ACT', '0')), p(out), pol._stream()))
    torch.cuda.synchronize()
    pr = pol.prof_read()["gemm"]
    if os.environ.get("STAMPS"):
        nblk = ((M + 255) // 256 + 7) // 8 * 8 * ((N + 255) // 256) if tile in (2, 4, 5, 6) or (tile == 0 and M >= 8192) else ((M + 127) // 128 + 7) // 8 * 8 * ((N + 127) // 128)
        dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
        pol.set_option("gemm_dbg_ptr", dbg.data_ptr())
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), p(bias), None, p(res), M, N, K, int(os.environ.get('ACT', '0')), p(out), pol._stream()))
        torch.cuda.synchronize()
        pol.set_option("gemm_dbg_ptr", 0)
        raw = dbg.view(-1, 8).cpu()
        raw = raw[raw[:, 3] > 0]
        d = raw[:, :4].double()
        t = d - d[:, :1]
        print(f"  stamps over {d.shape[0]} workgroups (shader clocks): prologue {t[:, 1].mean():.0f}  main loop {(t[:, 2] - t[:, 1]).mean():.0f}  "
              f"epilogue {(t[:, 3] - t[:, 2]).mean():.0f}  total {t[:, 3].mean():.0f}")
        # real time (100 MHz ticks) and placement: effective shader clock, and idle gaps between consecutive workgroups of one CU
        rt0, rt1 = raw[:, 4].double(), raw[:, 5].double()
        dur_us = (rt1 - rt0) / 100.0
        ghz = (t[:, 3] / (dur_us * 1e3))
        hw, xcc = raw[:, 6], raw[:, 7] & 0xF
        cu = (hw >> 8) & 0xF
        se = (hw >> 13) & 0x7
        sh = (hw >> 12) & 0x1
        key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        gaps, busy = [], []
        for k in key.unique():
            sel = (key == k).nonzero().flatten()
            o = sel[torch.argsort(rt0[sel])]
            if o.numel() > 1:
                gaps.append((rt0[o][1:] - rt1[o][:-1]) / 100.0)
            busy.append(float(dur_us[o].sum() / ((rt1[o].max() - rt0[o].min()) / 100.0)))
        g = torch.cat(gaps) if gaps else torch.zeros(1)
        print(f"  real time: {dur_us.mean():.2f} us per workgroup, effective shader clock {ghz.mean():.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f}); "
              f"{key.unique().numel()} distinct CUs, gap between consecutive workgroups of a CU mean {g.mean():.2f} us "
              f"(median {g.median():.2f}, p95 {g.quantile(0.95):.2f}); CU busy fraction {sum(busy) / len(busy):.3f}; "
              f"span {(rt1.max() - rt0.min()) / 100.0:.1f} us")
    ms = pr["ms"] / max(pr["launches"], 1)
    print(f"M{M} N{N} K{K} tile{tile} raster{os.environ.get('RASTER', '0')} epi{os.environ.get('EPI', '1')}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
