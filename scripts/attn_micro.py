"""T5 attention microbenchmark through the C ABI (vima_op_attention): python scripts/attn_micro.py B H L D [iters]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402


def main():
    B, H, L, D = (int(x) for x in sys.argv[1:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    mode = int(os.environ.get("MODE", "0"))
    Lq = int(os.environ.get("LQ", str(L)))
    pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
    pol._ensure_handle()
    pol.set_option("attn_qg", int(os.environ.get("QG", "1")))   # 2: 64 queries per wave
    q = torch.randn(B, Lq, H, D, device="cuda") * 0.4
    k = torch.randn(B, L, H, D, device="cuda") * 0.4
    v = torch.randn(B, L, H, D, device="cuda")
    mask = torch.ones(B, L, dtype=torch.bool, device="cuda")
    if os.environ.get("MASKF"):   # fraction of masked keys, scattered (the benchmark's prompts: 0.1); key 0 stays valid, batch row 1 is fully masked
        mask = torch.rand(B, L, device="cuda") >= float(os.environ["MASKF"])
        mask[:, 0] = True
        if B > 1:
            mask[1] = False
    rb = torch.randn(H, 2 * L - 1, device="cuda")
    out = torch.empty(B, Lq, H, D, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    scale = 1.0 if mode == 0 else D ** -0.5
    for _ in range(2):
        _lib.check(pol._lib.vima_op_attention(pol._handle, p(q), p(k), p(v), p(mask), p(rb), B, H, Lq, L, D, scale, mode, 1, p(out), pol._stream()))
    torch.cuda.synchronize()
    pol.prof_enable(True)
    for _ in range(iters):
        _lib.check(pol._lib.vima_op_attention(pol._handle, p(q), p(k), p(v), p(mask), p(rb), B, H, Lq, L, D, scale, mode, 1, p(out), pol._stream()))
    torch.cuda.synchronize()
    pr = pol.prof_read()["attention"]
    if os.environ.get("STAMPS"):   # needs a library built with -DVIMA_ATTN_STAMPS=1 (scripts/gpu_job_attn_ablate.sh pattern)
        nwg = B * H * ((Lq + 127) // 128) + 64
        dbg = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
        pol.set_option("attn_dbg_ptr", dbg.data_ptr())
        _lib.check(pol._lib.vima_op_attention(pol._handle, p(q), p(k), p(v), p(mask), p(rb), B, H, Lq, L, D, scale, mode, 1, p(out), pol._stream()))
        torch.cuda.synchronize()
        pol.set_option("attn_dbg_ptr", 0)
        d = dbg.view(-1, 8).cpu().double()
        d = d[d[:, 1] > 0][:, :6]
        nt = (L + 63) // 64
        names = ["issue global loads", "S^T (K frags + MFMA)", "softmax", "PV (V frags + MFMA)", "LDS store of next tile", "barrier"]
        tot = d.sum(1).mean()
        print("  wave-0 shader clocks per key tile (%d workgroups, %d tiles each): " % (d.shape[0], nt) +
              ", ".join(f"{n} {d[:, i].mean() / nt:.0f}" for i, n in enumerate(names)) + f"; total {tot / nt:.0f}")
    if os.environ.get("CHECK"):   # against the exact generic kernel
        ref = torch.empty_like(out)
        _lib.check(pol._lib.vima_op_attention(pol._handle, p(q), p(k), p(v), p(mask), p(rb), B, H, Lq, L, D, scale, mode, 0, p(ref), pol._stream()))
        torch.cuda.synchronize()
        print("  max |mfma - generic| =", (out - ref).abs().max().item(), "of", ref.abs().max().item())
    ms = pr["ms"] / max(pr["launches"], 1)
    print(f"attn mode{mode} B{B} H{H} Lq{Lq} Lk{L} D{D}: {ms:.3f} ms = {4.0 * B * H * Lq * L * D / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
