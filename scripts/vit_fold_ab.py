"""ViT LayerNorm fold (option vit_ln_fuse) A/B on ONE 16384-crop chunk (81920 token rows, the headline's chunk shape): per-shape GEMM time from the
library's HIP-event records with the fold off / on, and the `other` (LayerNorm, attention, embed ...) time. python scripts/vit_fold_ab.py [fuse ...]
(the T5 stack of the same call -- 64 x 512 rows -- is listed too: M = 32768)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_testing import synthetic as syn      # noqa: E402
from vima_amd.policy import VIMAPolicy         # noqa: E402

cfg = syn.config("2M", xattn_n_positions=512)
sd = syn.make_state_dict(cfg, 3)
prompts = syn.to_device(syn.make_prompt(64, n_segments=32, words_per_segment=8, q_per_view=4, seed=5), "cuda:0")
for fuse in ([int(a) for a in sys.argv[1:]] or [0, 1, 0, 1]):
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=cfg.xattn_n_positions, precision="bf16", device="cuda:0")
    pol.load_state_dict(sd)
    pol.set_option("dual_stream", 0)
    try:
        pol.set_option("vit_ln_fuse", fuse)
    except Exception:   # noqa: BLE001  (a library without the option: VIMA_HIP_LIB A/B against an older build)
        pass
    for _ in range(3):
        pol.forward_prompt_assembly(prompts)
    torch.cuda.synchronize()
    pol.prof_enable(True)
    for _ in range(5):
        pol.forward_prompt_assembly(prompts)
    torch.cuda.synchronize()
    by = collections.defaultdict(list)
    for r in pol.prof_read_gemm_launches():
        if r["M"] >= 16384:
            by[(r["kernel"], r["M"], r["N"], r["K"])].append(r["us"])
    pr = pol.prof_read()
    pol.prof_enable(False)
    print(f"vit_ln_fuse={fuse}: gemm {pr['gemm']['ms'] / 5:.3f} ms, other {pr['other']['ms'] / 5:.3f} ms ({pr['other']['launches'] // 5} launches) per call")
    for k, v in sorted(by.items()):
        print(f"   {k[0]:34s} M{k[1]:6d} N{k[2]:5d} K{k[3]:5d}: {len(v) // 5:2d} x {sum(v) / len(v):7.2f} us")
