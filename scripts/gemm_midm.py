"""Mid-size-M GEMM shapes of one decoder step at batch 256 (M = 2304) through the C ABI, per forced tile shape:
python scripts/gemm_midm.py [M]   (gemm_tile: 0 auto, 1 128x128, 8 64x64, 9 256x128, 2 256x256)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
pol._ensure_handle()
pol.set_option("op_bf16_out", 1)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
for (N, K, act) in ((768, 768, 0), (1536, 768, 0), (2304, 768, 0), (768, 3072, 0)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    out = torch.empty(M, N, device="cuda")
    ref = None
    for tile in (0, 1, 8, 13, 14):
        pol.set_option("gemm_tile", tile)
        try:
            for _ in range(3):
                _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
        except Exception as e:   # noqa: BLE001
            print(f"M{M} N{N} K{K} act{act} tile {tile}: {str(e)[:60]}")
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = torch.equal(out, ref)
        pol.prof_enable(True)
        for _ in range(20):
            _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
        torch.cuda.synchronize()
        pr = pol.prof_read()["gemm"]
        pol.prof_enable(False)
        us = pr["ms"] / max(pr["launches"], 1) * 1e3
        print(f"M{M} N{N} K{K} act{act} tile {tile}: {us:7.2f} us = {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  bit-identical to auto: {same}")
