"""Vendor-library yardstick (NOT part of the product path): torch.matmul (hipBLASLt) on the big VIMA-200M GEMM shapes,
bf16 in / bf16 out, timed with HIP events. python scripts/blaslt_ref.py"""
import torch

SHAPES = [(131072, 2304, 768), (131072, 768, 3072), (81920, 3072, 768), (131072, 768, 768), (131072, 3072, 768)]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    for _ in range(5):
        out = A @ W.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = A @ W.T
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"hipBLASLt M{M} N{N} K{K}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
