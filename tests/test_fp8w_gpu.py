"""precision="fp8w" (BASELINE.json configs[4]: fp8 weights): OCP E4M3 weights with per-output-channel scales for the large
Linear layers, widened to bf16 in registers by the GEMM kernels; bf16 activations / MFMA, fp32 accumulation.
Two questions, two gates:
  * is the fp8 kernel path correct?  -> against the oracle run on the SAME fake-quantised weights (oracle/fp8_quant.py):
    like the bf16 mode, raw logits within 1e-3 abs, tokens within 4 % of their maximum;
  * what do 3-bit-mantissa weights cost?  -> measured against the unmodified reference's goldens and REPORTED (the 1e-3
    north_star gate belongs to the bf16 mode); sanity-gated at 10 % of max|logit|."""
import os

import numpy as np
import pytest
import torch

from oracle.cases import CASES, build_case, run_policy, case_state_dict
from oracle.fp8_quant import fake_quant_state_dict
from oracle.vima_oracle import OraclePolicy, ACTION_KEYS
from vima_testing import synthetic as syn
from tests.gpu_common import bare_policy, loaded_policy, max_abs, max_rel, ptr
from tests.test_policy_gpu import native_outputs, _flip_report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name,fused", [("ragged_4M", 1), ("e384_long", 1), ("e384_long", 0), ("bench_200M", 1)])
def test_fp8w_matches_oracle_on_fake_quantised_weights(name, fused):
    cfg, wseed, prompts, obs, actions = build_case(name)
    sd = case_state_dict(name, cfg)
    orc = OraclePolicy(fake_quant_state_dict(sd, t5_fused_rms=bool(fused)), **cfg.ctor_kwargs())
    o, od = run_policy(orc, prompts, obs, actions)
    ref_logits = torch.cat([od[k]["raw"] for k in ACTION_KEYS], dim=-1)
    pol = loaded_policy(cfg, sd, "fp8w", t5_fuse_rms=fused)
    out = native_outputs(pol, prompts, obs, actions)
    for k in ("prompt_tokens", "obs_tokens", "predicted"):
        assert max_rel(out[k], o[k]) < 4e-2, (k, max_rel(out[k], o[k]))
    err = max_abs(out["raw_logits"], ref_logits)
    print(f"[fp8w] {name} fused={fused}: vs oracle on the same fp8 weights: max|logit err| {err:.3e} (max|logit| {ref_logits.abs().max():.3g})")
    assert err < 1e-3, err


@pytest.mark.parametrize("name", ["bench_200M", "cfg2_20M", "lp1024_200M"])
def test_fp8w_error_against_reference_goldens_is_reported(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg, wseed, prompts, obs, actions = build_case(name)
    sd = case_state_dict(name, cfg)
    pol = loaded_policy(cfg, sd, "fp8w")
    out = native_outputs(pol, prompts, obs, actions)
    ref = torch.from_numpy(gold["raw_logits"]).reshape(-1, 700)
    got = out["raw_logits"].cpu().reshape(-1, 700)
    err = max_abs(got, ref)
    agree, total, gap = _flip_report(got, ref)
    print(f"[fp8w] {name}: vs the fp32 reference: max|logit err| {err:.3e} of max|logit| {ref.abs().max():.3g} "
          f"({err / ref.abs().max().item():.2%}), argmax agreement {agree}/{total}")
    assert torch.isfinite(got).all() and err < 0.1 * ref.abs().max().item()


def test_fp8w_weight_bytes_and_small_layers_stay_bf16():
    """The packed replica is roughly half the bf16 one (the word-embedding table and norms stay fp32)."""
    cfg = syn.config("20M")
    sd = syn.make_state_dict(cfg, 0)
    free0 = torch.cuda.mem_get_info()[0]
    p16 = loaded_policy(cfg, sd, "bf16")
    free1 = torch.cuda.mem_get_info()[0]
    p8 = loaded_policy(cfg, sd, "fp8w")
    free2 = torch.cuda.mem_get_info()[0]
    b16, b8 = free0 - free1, free1 - free2
    table = 32128 * 768 * 4
    assert b8 < b16 and (b8 - table) < 0.62 * (b16 - table), (b16, b8)


def test_fp8w_incremental_decoding_and_kv_cache():
    """The episode-cache entry points (vima_decode_step, prompt K/V cache) in the fp8w precision: step-by-step decoding
    reproduces the re-fed history, cached and stateless forward agree bit for bit."""
    cfg = syn.config("20M")
    sd = syn.make_state_dict(cfg, 11)
    pol = loaded_policy(cfg, sd, "fp8w")
    ref = loaded_policy(cfg, sd, "fp8w")
    ref.cache_prompt_kv = False
    g = torch.Generator().manual_seed(5)
    B, Lp, Q, E, T = 3, 24, 8, cfg.embed_dim, 4
    ptok = torch.randn(Lp, B, E, generator=g).to(DEV)
    pmask = torch.ones(B, Lp, dtype=torch.bool)
    pmask[1, 17:] = False
    otok = torch.randn(T, B, Q, E, generator=g).to(DEV)
    omask = torch.ones(T, B, Q, dtype=torch.bool)
    atok = torch.randn(T - 1, B, E, generator=g).to(DEV)
    full = pol.forward(otok, omask.to(DEV), atok, ptok, pmask.to(DEV))
    assert torch.equal(pol.forward(otok, omask.to(DEV), atok, ptok, pmask.to(DEV)), full)          # cached K/V
    assert torch.equal(ref.forward(otok, omask.to(DEV), atok, ptok, pmask.to(DEV)), full)          # stateless
    for t in range(T):
        step = pol.forward_step(otok[t], omask[t], atok[t - 1] if t > 0 else None, ptok, pmask, t)
        assert max_rel(step, full[t]) < 4e-2, (t, max_rel(step, full[t]))
