"""precision="fp8" (BASELINE.json configs[4], VERDICT r2 missing item 1): fp8 e4m3 WEIGHTS AND ACTIVATIONS into the T5 stack's
GEMMs on the CDNA4 fp8 matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, gemm_pp_kernel<.., F8>): per-output-channel weight
scales, one static activation scale per (layer, site) calibrated by the handle's first pass.
  * is the fp8 path correct?  -> against the oracle run on the SAME fake-quantised weights AND activations (oracle/fp8_quant.py,
    scales read back from the handle);
  * what does it cost in accuracy? -> reported against the fp32 oracle;
  * did the fp8 kernel actually run? -> the per-kernel profile must name gemm_pp_kernel<.., true>."""
import pytest
import torch

from oracle.fp8_quant import make_fp8_act_oracle
from oracle.vima_oracle import OraclePolicy
from vima_testing import synthetic as syn
from tests.gpu_common import loaded_policy, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case():
    cfg = syn.config("2M", xattn_n_positions=512)
    sd = syn.make_state_dict(cfg, 21, head_gain=0.5)
    B = 28                                               # 28 x 512 = 14336 rows: 168 tiles of 256x256 in the N = 768 GEMMs
    prompts = syn.make_prompt(B, n_segments=64, words_per_segment=4, q_per_view=2, seed=501)
    obs = syn.make_obs(1, B, 2, seed=502)
    return cfg, sd, prompts, obs


def _logits(pol, ptok, pmask, obs):
    ot, om = pol.forward_obs_token(obs)
    return pol.action_logits(pol.forward(obs_token=ot, obs_mask=om, action_token=None, prompt_token=ptok, prompt_token_mask=pmask)[-1])


def test_fp8_activation_path_matches_fake_quant_oracle():
    cfg, sd, prompts, obs = _case()
    pol = loaded_policy(cfg, sd, "fp8", dual_stream=0)
    p, o = syn.to_device(prompts, DEV), syn.to_device(obs, DEV)
    assert pol.fp8_act_scales() is None
    ptok_cal, pmask = pol.forward_prompt_assembly(p)      # calibrating pass (fp8w kernels + max |x| per site)
    scales = pol.fp8_act_scales()
    assert scales is not None and tuple(scales.shape) == (12, 4) and bool((scales > 0).all())
    pol.prof_enable(True)
    ptok, pmask = pol.forward_prompt_assembly(p)          # fp8 activations
    torch.cuda.synchronize()
    kernels = pol.prof_read_gemm_kernels()
    pol.prof_enable(False)
    f8 = {k: v for k, v in kernels.items() if k.endswith(", true>")}
    assert sum(v["launches"] for v in f8.values()) == 48, kernels.keys()      # 12 layers x (qkv, o, wi, wo)
    lg = _logits(pol, ptok, pmask, o)
    idx = [0, 5, 27]
    cp, co = syn.cut_prompt(prompts, idx), syn.cut_obs(obs, idx)
    with torch.no_grad():
        orc8 = make_fp8_act_oracle(sd, scales, **cfg.ctor_kwargs())
        r_ptok, r_pmask = orc8.forward_prompt_assembly(cp)
        rt, rm = orc8.forward_obs_token(co)
        r_lg = orc8.action_logits(orc8.forward(rt, rm, None, r_ptok, r_pmask)[-1])
        orc = OraclePolicy(sd, **cfg.ctor_kwargs())
        f_ptok, f_pmask = orc.forward_prompt_assembly(cp)
        ft, fm = orc.forward_obs_token(co)
        f_lg = orc.action_logits(orc.forward(ft, fm, None, f_ptok, f_pmask)[-1])
    got_ptok, got_lg = ptok[:, idx].float().cpu(), lg[idx].float().cpu()

    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    # e4m3 has a 6-12 % step: a bf16-level difference in a value BEFORE quantisation flips its code, so the library and an oracle
    # that quantises its own fp32 activations cannot agree element by element (unlike fp8w, whose quantised WEIGHTS are identical on
    # both sides). The gate is therefore statistical: the library's deviation from the fp32 oracle must equal the deviation the
    # fake-quant oracle has from the fp32 oracle (same quantisation points, same scales) -- no additional error source -- and the two
    # quantised runs must be closer to each other than either is to fp32.
    e_lib32, e_orc32, e_lib_orc = rms(got_ptok, f_ptok), rms(r_ptok, f_ptok), rms(got_ptok, r_ptok)
    l_lib32, l_orc32, l_lib_orc = rms(got_lg, f_lg), rms(r_lg, f_lg), rms(got_lg, r_lg)
    e_cal = rms(ptok_cal[:, idx].float().cpu(), f_ptok)
    print(f"[fp8] T5 stack with e4m3 activations + weights (48 launches of gemm_pp_kernel<.., true>), relative RMS deviations: prompt tokens "
          f"library vs fp32 oracle {e_lib32:.3e}, fake-quant oracle vs fp32 oracle {e_orc32:.3e}, library vs fake-quant oracle {e_lib_orc:.3e}; "
          f"logits (max|logit| {f_lg.abs().max():.3g}): {l_lib32:.3e} / {l_orc32:.3e} / {l_lib_orc:.3e}; max abs logit error vs fp32 "
          f"{max_abs(got_lg, f_lg):.3e}; the calibrating (fp8w) pass vs fp32: prompt tokens {e_cal:.3e}")
    assert torch.isfinite(got_lg).all()
    assert e_lib32 < 1.25 * e_orc32 + 5e-3 and l_lib32 < 1.5 * l_orc32 + 5e-3
    assert e_lib_orc < 0.9 * max(e_lib32, e_orc32) + 5e-3


def test_fp8_vit_and_kv_projection_match_fake_quant_oracle():
    """The ViT (a chunk of 16384 crops) and the decoder's prompt K/V projections with fp8 activations: same statistical gate."""
    cfg = syn.config("2M", xattn_n_positions=512)
    sd = syn.make_state_dict(cfg, 22, head_gain=0.5)
    B = 64                                               # 64 prompts x 64 images x 4 objects = 16384 crops; 64 x 512 prompt rows
    prompts = syn.make_prompt(B, n_segments=64, words_per_segment=4, q_per_view=2, seed=601)
    obs = syn.make_obs(1, B, 2, seed=602)
    pol = loaded_policy(cfg, sd, "fp8", dual_stream=0)
    p, o = syn.to_device(prompts, DEV), syn.to_device(obs, DEV)
    ptok, pmask = pol.forward_prompt_assembly(p)          # calibrates T5 + ViT
    _logits(pol, ptok, pmask, o)                          # calibrates the K/V projection
    vs, ks, ts = pol.fp8_act_scales("vit"), pol.fp8_act_scales("kv"), pol.fp8_act_scales("t5")
    assert vs is not None and tuple(vs.shape) == (4, 4) and ks is not None and ts is not None
    pol.prof_enable(True)
    ptok, pmask = pol.forward_prompt_assembly(p)
    lg = _logits(pol, ptok, pmask, o)
    torch.cuda.synchronize()
    kernels = pol.prof_read_gemm_kernels()
    pol.prof_enable(False)
    n8 = sum(v["launches"] for k, v in kernels.items() if k.endswith(", true>"))
    assert n8 == 48 + 17 + cfg.xf_n_layers, (n8, list(kernels))     # T5 + ViT (3 blocks x 4 + the cls-only block's 5) + one K/V projection per decoder layer
    idx = [0, 7, 63]
    cp, co = syn.cut_prompt(prompts, idx), syn.cut_obs(obs, idx)
    with torch.no_grad():
        orc8 = make_fp8_act_oracle(sd, ts, vit_scales=vs, kv_scale=float(ks[0]), **cfg.ctor_kwargs())
        orc8.fq_vit = True
        r_ptok, r_pmask = orc8.forward_prompt_assembly(cp)
        orc8.fq_vit = False
        rt, rm = orc8.forward_obs_token(co)
        r_lg = orc8.action_logits(orc8.forward(rt, rm, None, r_ptok, r_pmask)[-1])
        orc = OraclePolicy(sd, **cfg.ctor_kwargs())
        f_ptok, f_pmask = orc.forward_prompt_assembly(cp)
        ft, fm = orc.forward_obs_token(co)
        f_lg = orc.action_logits(orc.forward(ft, fm, None, f_ptok, f_pmask)[-1])

    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    got_ptok, got_lg = ptok[:, idx].float().cpu(), lg[idx].float().cpu()
    e = (rms(got_ptok, f_ptok), rms(r_ptok, f_ptok), rms(got_ptok, r_ptok))
    l = (rms(got_lg, f_lg), rms(r_lg, f_lg), rms(got_lg, r_lg))
    print(f"[fp8] T5 + ViT + K/V projection with e4m3 activations ({n8} fp8 GEMM launches), relative RMS deviations (library vs fp32 / fake-quant "
          f"oracle vs fp32 / library vs fake-quant oracle): prompt tokens {e[0]:.3e} / {e[1]:.3e} / {e[2]:.3e}; logits {l[0]:.3e} / {l[1]:.3e} / {l[2]:.3e}")
    assert torch.isfinite(got_lg).all()
    assert e[0] < 1.25 * e[1] + 5e-3 and l[0] < 1.5 * l[1] + 5e-3


def test_fp8_small_batches_fall_back_to_fp8w_kernels():
    """Shapes the fp8 kernel does not cover (fewer than 160 full 256x256 tiles) keep bf16 activations: same results as fp8w."""
    cfg = syn.config("2M")
    sd = syn.make_state_dict(cfg, 3)
    prompts = syn.to_device(syn.make_prompt(2, n_segments=4, words_per_segment=4, q_per_view=2, seed=9), DEV)
    a = loaded_policy(cfg, sd, "fp8").forward_prompt_assembly(prompts)[0]
    b = loaded_policy(cfg, sd, "fp8w").forward_prompt_assembly(prompts)[0]
    assert torch.equal(a, b)


def test_fp8_keeps_bf16_activations_when_the_options_leave_the_fp8_kernel_unreachable():
    """the fp8-activation GEMM exists on the ping-pong persistent kernel only. With a documented option that takes the large
    GEMMs elsewhere (gemm_persist = 0, gemm_tile = 1, gemm_raster = 1) the forward must keep bf16 activations (fp8w kernels), not fail."""
    cfg, sd, prompts, obs = _case()
    p = syn.to_device(prompts, DEV)
    ref = None
    for key, val in (("gemm_persist", 0), ("gemm_tile", 1), ("gemm_raster", 1)):
        pol = loaded_policy(cfg, sd, "fp8", dual_stream=0)
        pol.set_option(key, val)
        pol.forward_prompt_assembly(p)                    # would calibrate if the stage were eligible
        assert pol.fp8_act_scales() is None, f"{key}={val}: the T5 stage must not be fp8-eligible"
        pol.prof_enable(True)
        ptok, pmask = pol.forward_prompt_assembly(p)      # used to fail with hipErrorInvalidValue inside launch_gemm
        torch.cuda.synchronize()
        kernels = pol.prof_read_gemm_kernels()
        pol.prof_enable(False)
        assert not any(k.endswith(", true>") for k in kernels), kernels.keys()
        assert bool(torch.isfinite(ptok).all())
        ref = ptok if ref is None else ref
        assert max_abs(ptok, ref) < 1e-5                   # every tile shape accumulates K in the same order (fp8w path)


def test_fp8_calibration_has_headroom_and_refuses_non_finite_maxima():
    """scale = headroom x max |x| / 448 (option fp8_headroom_pct, default 125); a calibrating batch with an inf activation
    is refused and leaves the handle uncalibrated."""
    cfg, sd, prompts, obs = _case()
    p = syn.to_device(prompts, DEV)
    pol = loaded_policy(cfg, sd, "fp8", dual_stream=0)
    pol.set_option("fp8_headroom_pct", 100)
    pol.forward_prompt_assembly(p)
    s100 = pol.fp8_act_scales().clone()
    pol.set_option("fp8_headroom_pct", 125)              # re-calibrates
    assert pol.fp8_act_scales() is None
    pol.forward_prompt_assembly(p)
    s125 = pol.fp8_act_scales()
    assert torch.allclose(s125, s100 * 1.25, rtol=1e-6)
    # poison: an infinite word embedding reaches the stream that enters the T5 stack
    sd_bad = {k: v.clone() for k, v in sd.items()}
    key = next(k for k in sd_bad if k.endswith("prompt_embedding._embed_layer.weight"))
    sd_bad[key][:] = float("inf")
    bad = loaded_policy(cfg, sd_bad, "fp8", dual_stream=0)
    with pytest.raises(Exception, match="non-finite"):
        bad.forward_prompt_assembly(p)
    assert bad.fp8_act_scales() is None


def test_fp8_benchmarked_shape_200m_lp1024_against_reference_golden(golden_dir):
    """`lp1024_fp8` in the bench record (VIMA-200M, 1024-token prompt, 81 % of the FLOPs on the fp8 MFMA) was never
    compared with the reference at that size WITH the fp8-activation kernels engaged -- the `lp1024_200M` golden is a batch of 2, which stays
    on the fp8w kernels (test_fp8_small_batches_fall_back_to_fp8w_kernels). Samples are independent, so the golden's two samples are rows
    0-1 of a batch of 64 here (65 536 prompt rows, 2 x 16 384 prompt crops: the T5 stack, the ViT chunks and the prompt K/V projections all
    clear the >= 160-tile bar, exactly like the benchmarked batch of 256); call 2 runs the fp8 kernels (asserted from the launch profile) and
    rows 0-1 are compared with what the UNMODIFIED reference computed (tests/golden/lp1024_200M.npz), gated against the error the fake-quant
    oracle -- same weights, same scales, same quantisation points -- has against that golden."""
    import os
    import numpy as np
    from oracle.cases import build_case, case_state_dict
    gold = np.load(os.path.join(golden_dir, "lp1024_200M.npz"))
    cfg, wseed, p_gold, o_gold, _ = build_case("lp1024_200M")
    sd = case_state_dict("lp1024_200M", cfg)
    B = 64
    p_fill = syn.make_prompt(B - 2, n_segments=64, words_per_segment=8, q_per_view=4, seed=7001)
    o_fill = syn.make_obs(1, B - 2, 4, seed=7002)
    prompts, obs = syn.concat_prompts(p_gold, p_fill), syn.concat_obs(o_gold, o_fill)
    pol = loaded_policy(cfg, sd, "fp8")                    # default options: the benchmarked path (dual_stream on)
    p, o = syn.to_device(prompts, DEV), syn.to_device(obs, DEV)
    ptok, pmask = pol.forward_prompt_assembly(p)           # call 1 calibrates T5 + ViT (fp8w kernels)
    _logits(pol, ptok, pmask, o)                           # ... and the K/V projection
    ts, vs, ks = pol.fp8_act_scales("t5"), pol.fp8_act_scales("vit"), pol.fp8_act_scales("kv")
    assert ts is not None and vs is not None and ks is not None, "the benchmarked shape must be fp8-eligible in all three groups"
    pol.prof_enable(True)
    ptok, pmask = pol.forward_prompt_assembly(p)           # call 2: fp8 activations
    pol.cache_prompt_kv = False                            # stateless decode: the K/V projections run (and are profiled) in this call
    lg = _logits(pol, ptok, pmask, o)
    torch.cuda.synchronize()
    kernels = pol.prof_read_gemm_kernels()
    pol.prof_enable(False)
    f8 = {k: v for k, v in kernels.items() if k.endswith(", true>")}
    n8 = sum(v["launches"] for v in f8.values())
    fl8, fl = sum(v["flops"] for v in f8.values()), sum(v["flops"] for v in kernels.values())
    # T5: 12 layers x (qkv, o, wi, wo) on each of the two batch halves (dual_stream) + two ViT chunks of 16 384 crops x 17 + one K/V projection per layer
    assert n8 == 2 * 48 + 2 * 17 + cfg.xf_n_layers, (n8, list(kernels))
    assert fl8 / fl > 0.75, fl8 / fl                                  # the record's "81 % of the FLOPs on the fp8 MFMA"
    ref_lg = torch.from_numpy(gold["raw_logits"])[0]                  # [2, 700]
    got_lg = lg[:2].float().cpu()
    assert torch.isfinite(lg).all()
    assert torch.equal(pmask[:2].cpu(), torch.from_numpy(gold["prompt_masks"]))
    with torch.no_grad():
        orc8 = make_fp8_act_oracle(sd, ts, vit_scales=vs, kv_scale=float(ks[0]), **cfg.ctor_kwargs())
        orc8.fq_vit = True
        r_ptok, r_pmask = orc8.forward_prompt_assembly(p_gold)
        orc8.fq_vit = False                                           # one observation's ViT keeps bf16 activations in the library
        rt, rm = orc8.forward_obs_token(o_gold)
        r_lg = orc8.action_logits(orc8.forward(rt, rm, None, r_ptok, r_pmask)[-1])

    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    e_lib, e_orc = max_abs(got_lg, ref_lg), max_abs(r_lg, ref_lg)
    r_lib, r_orc, r_cross = rms(got_lg, ref_lg), rms(r_lg, ref_lg), rms(got_lg, r_lg)
    from oracle.cases import gold_view
    t_lib = rms(gold_view("lp1024_200M", "prompt_tokens", ptok[:, :2]).float().cpu(), torch.from_numpy(gold["prompt_tokens"]))
    t_orc = rms(gold_view("lp1024_200M", "prompt_tokens", r_ptok), torch.from_numpy(gold["prompt_tokens"]))
    print(f"[fp8 @ VIMA-200M, Lp = 1024, batch {B}, {n8} launches of gemm_pp_kernel<.., true> = {fl8 / fl:.1%} of the GEMM FLOPs] rows 0-1 vs the "
          f"REFERENCE golden: max|logit err| library {e_lib:.3e}, fake-quant oracle {e_orc:.3e} (max|logit| {ref_lg.abs().max():.3g}); relative RMS "
          f"library {r_lib:.3e}, oracle {r_orc:.3e}, library vs oracle {r_cross:.3e}; prompt tokens relative RMS library {t_lib:.3e}, oracle {t_orc:.3e}")
    assert r_lib < 1.25 * r_orc + 5e-3, (r_lib, r_orc)
    assert t_lib < 1.25 * t_orc + 5e-3, (t_lib, t_orc)
    assert e_lib < 1.25 * e_orc + 1e-3, (e_lib, e_orc)
