"""Where the time of one gemm_resident_kernel launch goes (option gemm_dbg_ptr: shader-clock stamps per workgroup + 100-MHz real
time), the floor of the measurement (a one-workgroup, one-slice GEMM) and whether the shader clock is the issue (the same launches
right after a burst of large GEMMs). python scripts/small_m_stamps.py"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
pol._ensure_handle()
pol.set_option("op_bf16_out", 1)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731


def run(A, W, out, M, N, K, act, n):
    for _ in range(n):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))


def event_us(A, W, out, M, N, K, act, n=20):
    run(A, W, out, M, N, K, act, 3)
    torch.cuda.synchronize()
    pol.prof_enable(True)
    run(A, W, out, M, N, K, act, n)
    torch.cuda.synchronize()
    pr = pol.prof_read()["gemm"]
    pol.prof_enable(False)
    return pr["ms"] / max(pr["launches"], 1) * 1e3


big = (torch.randn(8192, 4096, device="cuda"), torch.randn(8192, 4096, device="cuda") * 0.02, torch.empty(8192, 8192, device="cuda"))
for (M, N, K, act, tile) in ((1, 32, 64, 0, 10), (9, 768, 768, 0, 0), (9, 768, 3072, 0, 0), (9, 3072, 768, 2, 0), (288, 768, 768, 0, 0), (512, 768, 3072, 0, 0)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    out = torch.empty(M, N, device="cuda")
    pol.set_option("gemm_resident", 1)
    pol.set_option("gemm_tile", tile)
    us = event_us(A, W, out, M, N, K, act)
    time.sleep(0.2)
    us_idle = event_us(A, W, out, M, N, K, act)
    run(*big, 8192, 8192, 4096, 0, 6)
    us_hot = event_us(A, W, out, M, N, K, act)
    nwg = 4096
    dbg = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    pol.set_option("gemm_dbg_ptr", dbg.data_ptr())
    run(A, W, out, M, N, K, act, 1)
    torch.cuda.synchronize()
    pol.set_option("gemm_dbg_ptr", 0)
    raw = dbg.view(-1, 8).cpu()
    raw = raw[raw[:, 3] > 0]
    d = raw.double()
    t0 = d[:, 0]
    dur_us = (d[:, 5] - d[:, 4]) / 100.0
    ghz = (d[:, 3] - t0) / (dur_us * 1e3)
    span = (d[:, 5].max() - d[:, 4].min()) / 100.0
    print(f"M{M} N{N} K{K} act{act}: events {us:.2f} us (after 0.2 s idle {us_idle:.2f}, right after 6 x 8192x8192x4096 {us_hot:.2f}); "
          f"{d.shape[0]} workgroups: issue {(d[:, 6] - t0).mean():.0f}  first chunk landed {(d[:, 1] - t0).mean():.0f}  main loop done "
          f"{(d[:, 2] - t0).mean():.0f}  end {(d[:, 3] - t0).mean():.0f} shader clocks; {dur_us.mean():.2f} us per workgroup, "
          f"clock {ghz.mean():.2f} GHz, first start to last end {span:.2f} us", flush=True)
    pol.set_option("gemm_resident", 0)
    pol.set_option("gemm_tile", 0)
    print(f"    ring tiles: events {event_us(A, W, out, M, N, K, act):.2f} us", flush=True)
