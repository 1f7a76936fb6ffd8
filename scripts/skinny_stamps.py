"""Where the time of one gemm_skinny_kernel launch goes (option gemm_dbg_ptr: shader-clock stamps per workgroup + 100-MHz real time):
python scripts/skinny_stamps.py   (VIMA_SKINNY_COLS=8|16|32 forces the tile width)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
pol._ensure_handle()
pol.set_option("op_bf16_out", 1)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731


def run(A, W, out, M, N, K, act, n, res=None):
    for _ in range(n):
        _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, p(res), M, N, K, act, p(out), pol._stream()))


def event_us(*a, n=30, **kw):
    run(*a, 3, **kw)
    torch.cuda.synchronize()
    pol.prof_enable(True)
    run(*a, n, **kw)
    torch.cuda.synchronize()
    pr = pol.prof_read()["gemm"]
    pol.prof_enable(False)
    return pr["ms"] / max(pr["launches"], 1) * 1e3


for (M, N, K, act) in ((1, 32, 64, 0), (9, 768, 768, 0), (9, 2304, 768, 0), (9, 768, 3072, 0), (9, 3072, 768, 2), (32, 768, 768, 0), (1, 6144, 768, 1)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    out = torch.empty(M, N, device="cuda")
    line = f"M{M} N{N} K{K} act{act}:"
    for sk in (1, 0):
        pol.set_option("gemm_skinny", sk)
        us = event_us(A, W, out, M, N, K, act)
        dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
        pol.set_option("gemm_dbg_ptr", dbg.data_ptr())
        run(A, W, out, M, N, K, act, 1)
        torch.cuda.synchronize()
        pol.set_option("gemm_dbg_ptr", 0)
        d = dbg.view(-1, 8).cpu().double()
        d = d[d[:, 3] > 0]
        t0 = d[:, 0]
        dur = (d[:, 5] - d[:, 4]) / 100.0
        span = (d[:, 5].max() - d[:, 4].min()) / 100.0
        ghz = (d[:, 3] - t0) / (dur * 1e3)
        name = "skinny" if sk else "resident"
        line += (f"\n    {name:8s} events {us:6.2f} us; {d.shape[0]:4d} workgroups: loads issued {(d[:, 6] - t0).mean():6.0f}  "
                 f"{'partials reduced' if sk else 'first chunk'} {(d[:, 1] - t0).mean():6.0f}  main loop done {(d[:, 2] - t0).mean():6.0f}  end {(d[:, 3] - t0).mean():6.0f} clocks; "
                 f"{dur.mean():.2f} us per workgroup at {ghz.mean():.2f} GHz, first start to last end {span:.2f} us")
    pol.set_option("gemm_skinny", 1)
    print(line, flush=True)
