"""Stage-by-stage parity + quick perf report on a real MI355X (run through gpurun). Writes gpurun_out/debug.log.
Uses the oracle only as the checker."""
from __future__ import annotations

import ctypes
import math
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vima_amd import _lib          # noqa: E402
from vima_testing import synthetic as syn
from vima_amd.policy import VIMAPolicy               # noqa: E402
from oracle.vima_oracle import OraclePolicy, ACTION_KEYS  # noqa: E402
from oracle.cases import build_case, run_policy      # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "debug.log"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item(), (a - b).abs().max().item()


def bf(x):
    return x.bfloat16().float()


def mk_policy(prec):
    p = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision=prec, device="cuda:0")
    p._ensure_handle()
    return p


def test_linear(pol, prec, variant):
    lib = pol._lib
    pol.set_option("gemm_variant", variant)
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 128, 64), (256, 256, 128), (100, 50, 512), (8, 768, 768), (333, 700, 256), (2048, 2304, 768), (130, 132, 1536)]
    for (M, N, K) in shapes:
        for (act, use_b, use_m, use_r) in [(0, 0, 0, 0), (1, 1, 0, 0), (2, 1, 1, 0), (3, 1, 0, 1), (0, 0, 0, 1)]:
            A = torch.randn(M, K, generator=g)
            W = torch.randn(N, K, generator=g) * K ** -0.5
            b = torch.randn(N, generator=g) if use_b else None
            m = torch.randn(M, N, generator=g) if use_m else None
            r = torch.randn(M, N, generator=g) if use_r else None
            if prec == "bf16":
                ref = bf(A) @ bf(W).T
            else:
                ref = A @ W.T
            if b is not None:
                ref = ref + b
            if act == 1:
                ref = torch.relu(ref)
            elif act == 2:
                ref = torch.nn.functional.gelu(ref)
            elif act == 3:
                ref = ref * torch.sigmoid(1.702 * ref)
            if m is not None:
                ref = ref * (bf(m) if prec == "bf16" else m)
            if r is not None:
                ref = ref + r
            d = [None if t is None else t.cuda() for t in (A, W, b, m, r)]
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act, ptr(out), pol._stream()))
            torch.cuda.synchronize()
            re, ae = relerr(out, ref)
            flag = "OK " if re < (2e-3 if prec == "bf16" else 2e-5) else "BAD"
            log(f"  linear[{prec},v{variant}] M{M} N{N} K{K} act{act} b{use_b} m{use_m} r{use_r}: rel {re:.2e} abs {ae:.2e} {flag}")


def test_layernorm(pol):
    lib = pol._lib
    g = torch.Generator().manual_seed(6)
    for rows, E, rms in [(7, 256, 0), (1000, 768, 0), (513, 384, 1), (64, 1024, 0)]:
        x = torch.randn(rows, E, generator=g) * 3 + 0.5
        ga = torch.randn(E, generator=g)
        be = torch.randn(E, generator=g)
        if rms:
            ref = ga * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
        else:
            ref = torch.nn.functional.layer_norm(x, (E,), ga, be, 1e-5)
        out = torch.empty(rows, E, device="cuda")
        xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()
        _lib.check(lib.vima_op_layernorm(pol._handle, ptr(xd), ptr(gd), None if rms else ptr(bd), 1e-6 if rms else 1e-5, rms, rows, E, ptr(out), pol._stream()))
        torch.cuda.synchronize()
        re, ae = relerr(out, ref)
        log(f"  layernorm rows{rows} E{E} rms{rms}: rel {re:.2e} abs {ae:.2e} {'OK ' if re < 1e-5 else 'BAD'}")


def attn_ref(q, k, v, kmask, relbias, scale, mode):
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q, k)
    FMIN = torch.finfo(torch.float32).min
    madd = (1.0 - kmask[:, None, None, :].float()) * FMIN if kmask is not None else 0.0
    if mode == 0:
        idx = (torch.arange(Lk)[None, :] - torch.arange(Lq)[:, None]) + Lk - 1
        s = s + (relbias[:, idx][None] + madd)
    elif mode == 1:
        s = s * scale + madd
    else:
        s = s * scale
        bmask = torch.tril(torch.ones(Lq, Lk))
        s = s * bmask + -1e4 * (1 - bmask)
        s = s + madd
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v)


def test_attention(pol, prec):
    lib = pol._lib
    g = torch.Generator().manual_seed(7)
    cfgs = [(0, 2, 12, 64, 64, 64), (0, 1, 12, 100, 100, 64), (1, 3, 8, 8, 40, 32), (1, 2, 24, 71, 512, 32), (2, 2, 8, 9, 9, 32),
            (2, 3, 24, 71, 71, 32), (1, 2, 4, 5, 33, 64), (2, 1, 4, 40, 40, 64)]
    for (mode, B, H, Lq, Lk, D) in cfgs:
        q = torch.randn(B, Lq, H, D, generator=g) * (1.0 if mode else 0.4)
        k = torch.randn(B, Lk, H, D, generator=g) * (1.0 if mode else 0.4)
        v = torch.randn(B, Lk, H, D, generator=g)
        kmask = torch.rand(B, Lk, generator=g) > 0.2
        kmask[:, 0] = True
        if B > 1 and mode == 1:
            kmask[1, :] = False   # fully masked row -> uniform (reference semantics)
        relbias = torch.randn(H, 2 * Lk - 1, generator=g) if mode == 0 else None
        scale = 1.0 if mode == 0 else 1.0 / math.sqrt(D)
        if prec == "bf16":
            ref = attn_ref(bf(q), bf(k), bf(v), kmask, relbias, scale, mode)
        else:
            ref = attn_ref(q, k, v, kmask, relbias, scale, mode)
        for impl in ([0, 1] if prec == "bf16" else [0]):
            out = torch.full((B, Lq, H, D), float("nan"), device="cuda")
            qd, kd, vd, md = q.cuda(), k.cuda(), v.cuda(), kmask.cuda()
            rd = relbias.cuda() if relbias is not None else None
            try:
                _lib.check(lib.vima_op_attention(pol._handle, ptr(qd), ptr(kd), ptr(vd), ptr(md), ptr(rd), B, H, Lq, Lk, D, scale, mode, impl, ptr(out), pol._stream()))
                torch.cuda.synchronize()
                re, ae = relerr(out, ref)
                tol = 2e-2 if prec == "bf16" else 1e-5
                log(f"  attention[{prec}] mode{mode} impl{impl} B{B} H{H} Lq{Lq} Lk{Lk} D{D}: rel {re:.2e} abs {ae:.2e} {'OK ' if re < tol else 'BAD'}")
            except Exception as e:
                log(f"  attention[{prec}] mode{mode} impl{impl}: EXC {e}")


def test_policy(name, prec, attn_impl=1, gemm_variant=1):
    cfg, wseed, prompts, obs, actions = build_case(name)
    sd = syn.make_state_dict(cfg, wseed)
    orc = OraclePolicy(sd, **cfg.ctor_kwargs())
    o_out, o_d = run_policy(orc, prompts, obs, actions)
    o_logits = torch.cat([o_d[k]["raw"] for k in ACTION_KEYS], dim=-1)
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=cfg.xattn_n_positions, precision=prec, device="cuda:0")
    pol.load_state_dict(sd, strict=True)
    pol.set_option("attn_impl", attn_impl)
    pol.set_option("gemm_variant", gemm_variant)
    dev = "cuda:0"
    p_d = syn.to_device(prompts, dev)
    obs_d = syn.to_device(obs, dev)
    act_d = syn.to_device(actions, dev) if actions is not None else None
    out, d = run_policy(pol, p_d, obs_d, act_d)
    torch.cuda.synchronize()
    logits = torch.cat([d[k].raw_logits for k in ACTION_KEYS], dim=-1)
    for k in o_out:
        if o_out[k].dtype == torch.bool:
            log(f"  policy[{name},{prec},a{attn_impl},g{gemm_variant}] {k}: equal={bool(torch.equal(out[k].cpu(), o_out[k]))}")
        else:
            re, ae = relerr(out[k], o_out[k])
            log(f"  policy[{name},{prec},a{attn_impl},g{gemm_variant}] {k}: rel {re:.2e} abs {ae:.2e}")
    re, ae = relerr(logits, o_logits)
    log(f"  policy[{name},{prec},a{attn_impl},g{gemm_variant}] raw_logits: rel {re:.2e} abs {ae:.2e} (max|logit| {o_logits.abs().max():.3g})")
    # stage isolation: feed ORACLE intermediates to each native stage
    ptok_o, pmask_o = o_out["prompt_tokens"].to(dev), o_out["prompt_masks"].to(dev)
    otok_o, omask_o = o_out["obs_tokens"].to(dev), o_out["obs_masks"].to(dev)
    atok_o = o_out["action_tokens"].to(dev) if "action_tokens" in o_out else None
    pred = pol.forward(otok_o, omask_o, atok_o, ptok_o, pmask_o)
    re, ae = relerr(pred, o_out["predicted"])
    log(f"  policy[{name},{prec}] decoder-only (oracle inputs): rel {re:.2e} abs {ae:.2e}")
    lg = pol.action_logits(o_out["predicted"][-1:].to(dev))
    re, ae = relerr(lg, o_logits)
    log(f"  policy[{name},{prec}] head-only (oracle inputs): rel {re:.2e} abs {ae:.2e}")
    del pol


def bench_gemm(pol, variant, tile=0):
    lib = pol._lib
    pol.set_option("gemm_variant", variant)
    pol.set_option("gemm_tile", tile)
    for (M, N, K) in [(131072, 2304, 768), (131072, 3072, 768), (131072, 768, 3072), (131072, 768, 768), (81920, 2304, 768),
                      (81920, 3072, 768), (131072, 1536, 768), (2048, 768, 768)]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.03
        out = torch.empty(M, N, device="cuda")
        for _ in range(2):
            _lib.check(lib.vima_op_linear(pol._handle, ptr(A), ptr(W), None, None, None, M, N, K, 0, ptr(out), pol._stream()))
        pol.prof_enable(True)
        for _ in range(5):
            _lib.check(lib.vima_op_linear(pol._handle, ptr(A), ptr(W), None, None, None, M, N, K, 0, ptr(out), pol._stream()))
        torch.cuda.synchronize()
        pr = pol.prof_read()
        pol.prof_enable(False)
        ms = pr["gemm"]["ms"] / max(pr["gemm"]["launches"], 1)
        log(f"  gemm-perf[{pol.precision},v{variant},tile{tile}] M{M} N{N} K{K}: {ms:.3f} ms/launch = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s (fp32-out epilogue)")


def main():
    log("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__)
    steps = sys.argv[1:] or ["ops", "policy", "perf"]
    if "ops" in steps:
        for prec in ("fp32", "bf16"):
            try:
                pol = mk_policy(prec)
                for variant in (1, 0):
                    test_linear(pol, prec, variant)
                test_layernorm(pol)
                test_attention(pol, prec)
                del pol
            except Exception:
                log("EXC in ops", prec, traceback.format_exc())
    if "policy" in steps:
        for name in ("cfg1_T2", "ragged_4M", "e384_long"):
            for prec, ai in (("fp32", 0), ("bf16", 0), ("bf16", 1)):
                try:
                    test_policy(name, prec, attn_impl=ai)
                except Exception:
                    log("EXC in policy", name, prec, ai, traceback.format_exc())
    if "perf" in steps:
        try:
            pol = mk_policy("bf16")
            for tile in (1, 2):
                bench_gemm(pol, 1, tile)
            del pol
        except Exception:
            log("EXC in perf", traceback.format_exc())
    log("done")


if __name__ == "__main__":
    main()
