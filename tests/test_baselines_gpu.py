"""Baseline policies (VIMAGPTPolicy / VIMAGatoPolicy / VIMAFlamingoPolicy) on a real MI355X, through
the C ABI and the host mirrors of vima_amd/baselines.py:
  * against golden fixtures produced by the UNMODIFIED reference modules (tests/golden/baseline_*.npz),
  * against the oracle on inputs that are not in the golden set.
fp32-operand mode: 1e-3 abs / 2e-4 rel on every stage tensor. bf16 mode: 4e-2 of max|.| on tokens; the action-head gain
of these cases is 0.5 (logits O(1)), so the logits are gated relative to max|logit| (2 %) and argmax agreement is reported."""
import os

import numpy as np
import pytest
import torch

from oracle.baseline_oracle import build_baseline_oracle
from oracle.cases import BASELINE_CASES, build_baseline_case, baseline_state_dict, run_baseline
from vima_amd import _lib
from vima_testing import synthetic as syn
from vima_amd.baselines import build_baseline
from tests.gpu_common import max_abs, max_rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _native(cfg, sd, prec, prompts, obs, actions):
    pol = build_baseline(cfg, precision=prec, device=DEV)
    pol.load_state_dict(sd, strict=True)
    out = run_baseline(pol, syn.to_device(prompts, DEV), syn.to_device(obs, DEV),
                       syn.to_device(actions, DEV) if actions is not None else None)
    out["raw_logits"] = pol.action_logits(out["predicted"][-1:])
    out["obj_encoder"] = pol.obj_encoder(syn.to_device(prompts[2]["rgb"], DEV))
    torch.cuda.synchronize()
    return pol, out


def _compare(out, ref_of, prec, tag):
    for k, ref in ref_of.items():
        got = out[k].cpu()
        assert tuple(got.shape) == tuple(ref.shape), (tag, k, tuple(got.shape), tuple(ref.shape))
        if ref.dtype == torch.bool:
            assert torch.equal(got, ref), (tag, k)
        elif prec == "fp32":
            assert max_abs(got, ref) < 1e-3 * max(1.0, ref.abs().max().item()), (tag, k, max_abs(got, ref))
            assert max_rel(got, ref) < 2e-4, (tag, k, max_rel(got, ref))
        elif k == "raw_logits":
            assert max_rel(got, ref) < 2e-2, (tag, k, max_rel(got, ref))
        else:
            assert max_rel(got, ref) < 4e-2, (tag, k, max_rel(got, ref))


@pytest.mark.parametrize("name", list(BASELINE_CASES))
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_baseline_matches_reference_golden(name, prec, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg, prompts, obs, actions = build_baseline_case(name)
    sd = baseline_state_dict(name, cfg)
    _, out = _native(cfg, sd, prec, prompts, obs, actions)
    _compare(out, {k: torch.from_numpy(gold[k]) for k in gold.files if not k.startswith("_")}, prec, name)
    if prec == "bf16":
        g, r = out["raw_logits"].cpu().reshape(-1, 700), torch.from_numpy(gold["raw_logits"]).reshape(-1, 700)
        agree = total = 0
        off = 0
        for k in syn.ACTION_KEYS:
            for bins in syn.ACTION_DIMS[k]:
                agree += int((g[:, off:off + bins].argmax(-1) == r[:, off:off + bins].argmax(-1)).sum())
                total += g.shape[0]
                off += bins
        print(f"[{name}] bf16 argmax agreement {agree}/{total}, logits rel err {max_rel(g, r):.2e}")
        assert agree >= 0.8 * total


@pytest.mark.parametrize("kind", ["gpt", "gato", "flamingo"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_baseline_matches_oracle_other_shapes(kind, prec):
    """Not in the golden set: E = 384 (12 heads of 32; Perceiver head dim 48), 3 layers, batch 5 with ragged prompts, T = 2
    with L_act = T (action token after the last observation), a word-only prompt in the batch."""
    cfg = syn.BaselineConfig(kind, 384, 3, 12, xattn_n_heads=12 if kind == "flamingo" else 0, vocab_size=8)
    sd = syn.make_baseline_state_dict(cfg, seed=41, head_gain=0.5)
    layout = [[0, 1, 0, 0], [0, 0, 1, 0, 1, 0, 0], [0, 0, 0], [1], [0, 1, 1, 0]]
    prompts = syn.make_rgb_prompt(5, layout=layout, seed=42)
    obs = syn.make_rgb_obs(2, 5, seed=43)
    actions = syn.make_actions(2, 5, seed=44)
    orc = build_baseline_oracle(cfg, sd)
    ref = run_baseline(orc, prompts, obs, actions)
    ref["raw_logits"] = orc.action_logits(ref["predicted"][-1:])
    ref["obj_encoder"] = orc.obj_encoder(prompts[2]["rgb"])
    pol, out = _native(cfg, sd, prec, prompts, obs, actions)
    _compare(out, ref, prec, kind)
    # batch-composition invariance of the decoder-only path: sample 1 alone (its own, shorter padding) gives the same rows
    if prec == "fp32":
        p1 = ([layout[1]], prompts[1][sum(t == 0 for t in layout[0]):sum(t == 0 for t in layout[0]) + 5],
              syn.MapDict(rgb=syn.MapDict({v: prompts[2]["rgb"][v][1:3] for v in syn.VIEWS})))
        o1 = syn.MapDict(rgb=syn.MapDict({v: obs["rgb"][v][:, 1:2] for v in syn.VIEWS}), ee=obs["ee"][:, 1:2])
        a1 = {k: v[:, 1:2] for k, v in actions.items()}
        one = run_baseline(pol, syn.to_device(p1, DEV), syn.to_device(o1, DEV), syn.to_device(a1, DEV))
        assert max_abs(one["predicted"][:, 0], out["predicted"][:, 1]) < 2e-4, max_abs(one["predicted"][:, 0], out["predicted"][:, 1])


def test_baseline_entry_points_reject_the_wrong_policy_kind():
    """The object-crop entry points of VIMAPolicy and the whole-frame ones of the baselines are not interchangeable: a
    handle of the other kind fails loudly instead of reading weights it does not have."""
    from tests.gpu_common import loaded_policy
    cfg = syn.BaselineConfig("gato", 256, 1, 8, vocab_size=8)
    pol = build_baseline(cfg, precision="bf16", device=DEV)
    pol.load_state_dict(syn.make_baseline_state_dict(cfg, 0), strict=True)
    obs = syn.make_obs(1, 2, 2, seed=1)
    with pytest.raises(_lib.VimaError):
        from vima_amd.policy import VIMAPolicy
        VIMAPolicy.forward_obs_token(pol, syn.to_device(obs, DEV))
    vcfg = syn.config("2M")
    vp = loaded_policy(vcfg, syn.make_state_dict(vcfg, 0), "bf16")
    lib = _lib.load()
    z = torch.zeros(4, 3, 64, 128, dtype=torch.uint8, device=DEV)
    out = torch.empty(4, 16, 256, device=DEV)
    import ctypes
    rc = lib.vima_rgb_encode(vp._handle, (ctypes.c_void_p * 2)(z.data_ptr(), z.data_ptr()), 4, ctypes.c_void_p(out.data_ptr()), None)
    assert rc != 0 and b"baseline" in lib.vima_last_error()
    # sequence longer than n_positions -> IndexError like the positions_embed lookup of the reference
    long_prompt = torch.zeros(500, 2, 256, device=DEV)
    mask = torch.ones(2, 500, dtype=torch.bool, device=DEV)
    with pytest.raises(IndexError):
        pol.forward(torch.zeros(1, 2, 16, 256, device=DEV), None, long_prompt, mask)
    # a sample without any valid prompt token: the reference builds position id -1 for it and its embedding lookup raises
    # (vima_gato_policy.py:156-170); the host mirror refuses the batch the same way (the C entry point is asynchronous and clamps)
    short = torch.zeros(8, 2, 256, device=DEV)
    m2 = torch.ones(2, 8, dtype=torch.bool, device=DEV)
    m2[1] = False
    with pytest.raises(IndexError):
        pol.forward(torch.zeros(1, 2, 16, 256, device=DEV), None, short, m2)


def test_flamingo_incremental_decoding_matches_full_history():
    """VIMAFlamingoPolicy decodes with XAttnGPT: `forward_step` (episode K/V caches in the native handle) reproduces the rows of
    the full-history `forward` that the reference's loop computes by re-feeding everything (fp32-operand mode, 2e-5)."""
    cfg = syn.BaselineConfig("flamingo", 256, 2, 8, xattn_n_heads=8)
    sd = syn.make_baseline_state_dict(cfg, seed=51)
    pol = build_baseline(cfg, precision="fp32", device=DEV)
    pol.load_state_dict(sd, strict=True)
    T, B = 3, 2
    prompts = syn.to_device(syn.make_rgb_prompt(B, layout=[[0, 1, 0], [1, 0, 0, 1]], seed=52), DEV)
    obs = syn.to_device(syn.make_rgb_obs(T, B, seed=53), DEV)
    actions = syn.to_device(syn.make_actions(T - 1, B, seed=54), DEV)
    with torch.no_grad():
        ptok, pmask = pol.forward_prompt_assembly(prompts)
        otok = pol.forward_obs_token(obs)
        atok = pol.forward_action_token(actions)
        full = pol.forward(otok, atok, ptok, pmask)
        for t in range(T):
            step = pol.forward_step(otok[t], None if t == 0 else atok[t - 1], ptok, pmask, t)
            assert max_abs(step, full[t]) < 2e-5 * max(1.0, full.abs().max().item()), (t, max_abs(step, full[t]))
    gato = build_baseline(syn.BaselineConfig("gato", 256, 1, 8, vocab_size=8), device=DEV)
    with pytest.raises(NotImplementedError):
        gato.forward_step(None, None, None, None, 0)
    with pytest.raises(ValueError):
        build_baseline(syn.BaselineConfig("gato", 256, 1, 8, vocab_size=8), precision="fp8w", device=DEV)
