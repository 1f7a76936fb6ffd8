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
    pol.set_option("gemm_raster", int(os.environ.get("RASTER", "0")))
    pol.set_option("gemm_epi", int(os.environ.get("EPI", "1")))
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    out = torch.empty(M, N, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(iters):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, 0, p(out), pol._stream()))
    torch.cuda.synchronize()
    pol.prof_enable(True)
    for _ in range(iters):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, 0, p(out), pol._stream()))
    torch.cuda.synchronize()
    pr = pol.prof_read()["gemm"]
    if os.environ.get("STAMPS"):
        nblk = ((M + 255) // 256 + 7) // 8 * 8 * ((N + 255) // 256) if tile == 2 else ((M + 127) // 128 + 7) // 8 * 8 * ((N + 127) // 128)
        dbg = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        pol.set_option("gemm_dbg_ptr", dbg.data_ptr())
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, 0, p(out), pol._stream()))
        torch.cuda.synchronize()
        pol.set_option("gemm_dbg_ptr", 0)
        d = dbg.view(-1, 4).cpu().double()
        d = d[d[:, 3] > 0]
        t = d - d[:, :1]
        print(f"  stamps over {d.shape[0]} workgroups (shader clocks): prologue {t[:, 1].mean():.0f}  main loop {(t[:, 2] - t[:, 1]).mean():.0f}  "
              f"epilogue {(t[:, 3] - t[:, 2]).mean():.0f}  total {t[:, 3].mean():.0f}; kernel span {(d[:, 3].max() - d[:, 0].min()):.0f}")
    ms = pr["ms"] / max(pr["launches"], 1)
    print(f"M{M} N{N} K{K} tile{tile} raster{os.environ.get('RASTER', '0')} epi{os.environ.get('EPI', '1')}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
