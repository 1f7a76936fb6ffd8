"""Small-M A/B on the GPU box: the same VIMA-200M policy with the underfilled GEMM grids on the 4-deep ring
tiles (gemm_resident=0) and on gemm_resident_kernel (=1, optionally with a larger grid limit), variants interleaved in one process:
  * per-shape GEMM micro-timings through vima_op_linear (HIP events of the library's profiler), outputs compared bitwise;
  * cold step at batch 1 and 32, WARM step (prompt K/V cached) and incremental env step at batch 256.
Usage: python scripts/small_m_ab.py [micro] [steps]"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib    # noqa: E402
from vima_testing import synthetic as syn
from vima_amd.policy import VIMAPolicy         # noqa: E402

DEV = torch.device("cuda", 0)
VARIANTS = [("ring", {"gemm_resident": 0, "gemm_res_maxwg": 256, "gemm_res_nch": 0}), ("resident", {"gemm_resident": 1, "gemm_res_maxwg": 256, "gemm_res_nch": 0}),
            # unmeasured at the end of round 3 (no GPU minutes left): 64-KiB rings = two workgroups per CU, so that the 432-workgroup grids of a
            # batch-256 env step (M = 2304) are co-resident like the ring tiles' while keeping one barrier per chunk
            ("resident, 2 buffers, grids <= 512", {"gemm_resident": 1, "gemm_res_maxwg": 512, "gemm_res_nch": 2})]


def micro():
    pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision="bf16", device="cuda:0")
    pol._ensure_handle()
    pol.set_option("op_bf16_out", 1)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    for M in (9, 36, 288, 512, 2304):
        for (N, K, act) in ((768, 768, 0), (2304, 768, 0), (3072, 768, 2), (768, 3072, 0)):
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda") * 0.03
            out = torch.empty(M, N, device="cuda")
            ref = None
            line = f"M{M:5d} N{N:5d} K{K:5d} act{act}:"
            for name, res, tile, nch in (("ring", 0, 0, 0), ("auto", 1, 0, 0), ("32x32", 1, 10, 0), ("64x32", 1, 11, 0), ("64x64", 1, 12, 0),
                                         ("32x32/5buf", 1, 10, 5), ("64x64/5buf", 1, 12, 5), ("64x64/2buf", 1, 12, 2)):   # 2 buffers = 64 KiB: two workgroups per CU
                if name.startswith("32x32") and M > 600:
                    continue
                pol.set_option("gemm_resident", res)
                pol.set_option("gemm_tile", tile)
                pol.set_option("gemm_res_nch", nch)
                for _ in range(3):
                    _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                same = torch.equal(out, ref)
                pol.prof_enable(True)
                for _ in range(20):
                    _lib.check(pol._lib.vima_op_linear(pol._handle, p(A), p(W), None, None, None, M, N, K, act, p(out), pol._stream()))
                torch.cuda.synchronize()
                kinds = "+".join(k.split("::")[-1] for k in pol.prof_read_gemm_kernels())
                pr = pol.prof_read()["gemm"]
                pol.prof_enable(False)
                us = pr["ms"] / max(pr["launches"], 1) * 1e3
                line += f"  {name} {us:6.2f} us{'' if same else ' DIFFERENT'}" + (f" [{kinds}]" if name == "auto" else "")
            print(line, flush=True)
    pol.set_option("gemm_tile", 0)
    pol.set_option("gemm_res_nch", 0)


def steps(n):
    cfg = syn.config("200M", xattn_n_positions=512)
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=512, precision="bf16", device=DEV)
    pol.load_state_dict(syn.make_state_dict(cfg, 0), strict=True)

    def timed(fn, reps, inner):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(inner):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / inner * 1e3)
        return best

    for B in (1, 32):
        prompts = syn.to_device(syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1236), DEV)
        obs = syn.to_device(syn.make_obs(1, B, 4, seed=1336), DEV)

        def cold():
            ptok, pmask = pol.forward_prompt_assembly(prompts)
            otok, omask = pol.forward_obs_token(obs)
            return pol.action_logits(pol.forward(otok, omask, None, ptok, pmask)[-1])

        ref = None
        for name, opts in VARIANTS:
            for k, v in opts.items():
                pol.set_option(k, v)
            cold(); out = cold()
            ref = out.clone() if ref is None else ref
            ms = timed(cold, 3, n)
            print(f"cold  B={B:3d} {name:22s} {ms:8.3f} ms  bit-identical logits: {torch.equal(out, ref)}", flush=True)
    B = 256
    prompts = syn.to_device(syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1236), DEV)
    obs = syn.to_device(syn.make_obs(1, B, 4, seed=1336), DEV)
    act1 = syn.to_device(syn.make_actions(1, B, seed=1636), DEV)
    ptok, pmask = pol.forward_prompt_assembly(prompts)

    def warm():
        otok, omask = pol.forward_obs_token(obs)
        return pol.action_logits(pol.forward(otok, omask, None, ptok, pmask)[-1])

    def episode():
        for t in range(1, 9):
            otok, omask = pol.forward_obs_token(obs)
            atok = pol.forward_action_token(act1)
            last = pol.action_logits(pol.forward_step(otok, omask, atok, ptok, pmask, t))
        return last

    ref = refi = None
    for name, opts in VARIANTS:
        for k, v in opts.items():
            pol.set_option(k, v)
        warm(); out = warm()
        ref = out.clone() if ref is None else ref
        ms = timed(warm, 3, n)
        otok, omask = pol.forward_obs_token(obs)
        pol.forward_step(otok, omask, None, ptok, pmask, 0)
        outi = episode()
        refi = outi.clone() if refi is None else refi
        pol.forward_step(otok, omask, None, ptok, pmask, 0)
        msi = timed(lambda: (pol.forward_step(otok, omask, None, ptok, pmask, 0), episode()), 2, 1) / 9
        print(f"warm  B=256 {name:22s} {ms:8.3f} ms  (bit-identical: {torch.equal(out, ref)})   incremental env step ~{msi:7.3f} ms "
              f"(bit-identical: {torch.equal(outi, refi)})", flush=True)


if __name__ == "__main__":
    if "micro" in sys.argv[1:]:
        micro()
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    if nums or "micro" not in sys.argv[1:]:
        steps(nums[0] if nums else 20)
