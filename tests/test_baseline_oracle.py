"""The baseline-policy oracle (oracle/baseline_oracle.py: VIMAGPTPolicy / VIMAGatoPolicy / VIMAFlamingoPolicy restated)
against golden vectors produced by the UNMODIFIED reference modules (oracle/make_golden.py::main_baselines), plus a live
cross-check on inputs that are not in the golden set when /root/reference is present. CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_shim
from oracle.baseline_oracle import build_baseline_oracle
from oracle.cases import BASELINE_CASES, build_baseline_case, baseline_state_dict, run_baseline
from vima_testing import synthetic as syn

ATOL = 5e-5


@pytest.mark.parametrize("name", list(BASELINE_CASES))
def test_baseline_oracle_matches_reference_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg, prompts, obs, actions = build_baseline_case(name)
    sd = baseline_state_dict(name, cfg)
    assert abs(syn.state_dict_checksum(sd) - float(gold["_sd_checksum"])) <= 1e-6 * float(gold["_sd_checksum"])
    orc = build_baseline_oracle(cfg, sd)
    out = run_baseline(orc, prompts, obs, actions)
    out["raw_logits"] = orc.action_logits(out["predicted"][-1:])
    out["obj_encoder"] = orc.obj_encoder(prompts[2]["rgb"])
    for k in gold.files:
        if k.startswith("_"):
            continue
        ref = torch.from_numpy(gold[k])
        assert tuple(out[k].shape) == tuple(ref.shape), k
        if ref.dtype == torch.bool:
            assert torch.equal(out[k], ref), k
        else:
            err = (out[k] - ref).abs().max().item()
            assert err <= ATOL * max(1.0, ref.abs().max().item()), f"{name}/{k}: max abs err {err}"


@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")
@pytest.mark.parametrize("kind", ["gpt", "gato", "flamingo"])
def test_live_reference_matches_baseline_oracle(kind):
    """E = 320 (10 heads of 32), 3 layers, T = 2 with one and with two action tokens, padded prompts."""
    cfg = syn.BaselineConfig(kind, 320, 3, 10, xattn_n_heads=10 if kind == "flamingo" else 0, vocab_size=8)
    sd = syn.make_baseline_state_dict(cfg, seed=21)
    pol = ref_shim.build_reference_baseline(kind, **cfg.ctor_kwargs())
    pol.load_state_dict(sd, strict=True)
    orc = build_baseline_oracle(cfg, sd)
    prompts = syn.make_rgb_prompt(2, layout=[[0, 1, 0, 0], [0, 0, 1, 0, 1, 0, 0]], seed=31)
    obs = syn.make_rgb_obs(2, 2, seed=32)
    for n_act in (1, 2):
        actions = syn.make_actions(n_act, 2, seed=33)
        ref = run_baseline(pol, prompts, obs, actions)
        out = run_baseline(orc, prompts, obs, actions)
        for k in ref:
            if ref[k].dtype == torch.bool:
                assert torch.equal(ref[k], out[k]), k
            else:
                assert (ref[k] - out[k]).abs().max().item() < 5e-5 * max(1.0, ref[k].abs().max().item()), (k, n_act)
