"""256x384 persistent GEMM (option gemm_wide) against the 256x256 kernels on the same inputs (bit-exact expected: both
accumulate K in the same order), plus timing. python scripts/wide_check.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def run(pol, A, W, bias, res, act, out):
    M, K = A.shape
    N = W.shape[0]
    _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), p(bias), None, p(res), M, N, K, act, p(out), pol._stream()))


def main():
    pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
    pol._ensure_handle()
    pol.set_option("op_bf16_out", 1)
    g = torch.Generator(device="cuda").manual_seed(0)
    ok = True
    for (M, N, K, act, use_bias, stream) in [(2048 * 16, 768, 768, 0, True, True), (2048 * 16, 2304, 768, 0, True, False), (2048 * 16, 3072, 768, 3, True, False),
                                             (2048 * 16, 768, 3072, 0, False, True), (2048 * 16, 3072, 768, 1, False, False), (256 * 64, 1536, 768, 0, False, False),
                                             (131072, 2304, 768, 0, False, False), (131072, 768, 3072, 0, True, True), (131072, 3072, 768, 1, False, False)]:
        A = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(N, K, device="cuda", generator=g) * 0.03
        bias = torch.randn(N, device="cuda", generator=g) if use_bias else None
        res = torch.randn(M, N, device="cuda", generator=g) if stream else None
        pol.set_option("op_stream_T", 1 if stream else 0)
        outs, ms = [], []
        for wide in (0, 1):
            pol.set_option("gemm_wide", wide)
            out = torch.empty(M, N, device="cuda")
            for _ in range(2):
                run(pol, A, W, bias, res, act, out)
            torch.cuda.synchronize()
            pol.prof_enable(True)
            for _ in range(5):
                run(pol, A, W, bias, res, act, out)
            torch.cuda.synchronize()
            pr = pol.prof_read_ex()
            t = pr["gemm"]["ms"] + pr["gemm_residual"]["ms"]
            n = pr["gemm"]["launches"] + pr["gemm_residual"]["launches"]
            pol.prof_enable(False)
            outs.append(out)
            ms.append(t / max(n, 1))
        d = (outs[0] - outs[1]).abs().max().item()
        ok = ok and d == 0.0
        print(f"M{M} N{N} K{K} act{act} bias{int(use_bias)} stream{int(stream)}: 256x256 {ms[0]:.3f} ms ({2.0 * M * N * K / ms[0] / 1e9:.0f} TF)  "
              f"256x384 {ms[1]:.3f} ms ({2.0 * M * N * K / ms[1] / 1e9:.0f} TF)  max|diff| {d:.3e}")
    print("ALL BIT-EXACT" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
