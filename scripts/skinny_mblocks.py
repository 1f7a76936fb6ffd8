"""gemm_skinny_kernel on 32-row blocks of taller problems (option gemm_skinny_maxm) against the tiles the dispatcher picks otherwise, through the
C ABI: python scripts/skinny_mblocks.py   (per shape: us per launch with maxm = 32 (off) / 4096 (on), max |diff| between the two)"""
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
for M in (48, 64, 80, 160, 288, 512, 1024, 2048):
    for (N, K, act) in ((768, 768, 0), (2304, 768, 0), (3072, 768, 2), (768, 3072, 0)):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.03
        outs, us = [], []
        for maxm in (32, 4096):
            pol.set_option("gemm_skinny_maxm", maxm)
            out = torch.empty(M, N, device="cuda")
            for _ in range(3):
                _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
            torch.cuda.synchronize()
            pol.prof_enable(True)
            for _ in range(30):
                _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
            torch.cuda.synchronize()
            pr = pol.prof_read()["gemm"]
            pol.prof_enable(False)
            us.append(pr["ms"] / max(pr["launches"], 1) * 1e3)
            outs.append(out)
        d = (outs[0] - outs[1]).abs().max().item()
        print(f"M{M:5d} N{N:5d} K{K:5d} act{act}: tiles {us[0]:6.2f} us  skinny {us[1]:6.2f} us  ({us[0] / us[1]:4.2f}x)  max|diff| {d:.2e}")
