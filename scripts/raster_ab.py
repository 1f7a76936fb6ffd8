"""One large K = 768 GEMM through the C ABI (vima_op_linear, bf16 output) for the raster A/B of VERDICT r4 item 4:
python scripts/raster_ab.py M N K act [launches]   with VIMA_GEMM_NGROUP_KB in the environment (2560 = shipped: the n-tiles of an XCD
in groups whose W panels fit its L2, the A panels re-read once per group; 1000000 = ONE group: an XCD walks all n-tiles of an A panel
before it moves on, A is fetched once). Prints the median / best launch time from HIP events."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib                      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

M, N, K, act = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 21
pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
pol._ensure_handle()
pol.set_option("op_bf16_out", 1)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.rand(M, K, device="cuda", generator=g) * 2 - 1
W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
torch.cuda.synchronize()
pol.prof_enable(True)
for _ in range(n):
    _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
torch.cuda.synchronize()
ls = pol.prof_read_gemm_launches()
pol.prof_enable(False)
us = sorted(l["us"] for l in ls)
name = ls[0]["kernel"]
med, best = us[len(us) // 2], us[0]
print(f"M{M} N{N} K{K} act{act} ngroup_kb={os.environ.get('VIMA_GEMM_NGROUP_KB', '2560 (default)')} [{name}]: median {med:.1f} us = "
      f"{2.0 * M * N * K / med / 1e6:.1f} TFLOP/s ; best {best:.1f} us = {2.0 * M * N * K / best / 1e6:.1f} TFLOP/s over {len(us)} launches")
