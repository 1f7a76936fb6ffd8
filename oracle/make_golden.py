"""TEST INFRASTRUCTURE: generate tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/vima, imported through oracle/ref_shim.py) on the seeded synthetic weights/inputs of
oracle/cases.py. Run in the build container only:  python -m oracle.make_golden
Also prints the oracle-vs-reference deviation for every stored tensor."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim                       # noqa: E402
from oracle.cases import (CASES, BASELINE_CASES, build_case, run_policy, case_state_dict, gold_view, gold_keys,  # noqa: E402
                          build_baseline_case, baseline_state_dict, run_baseline)
from oracle.baseline_oracle import build_baseline_oracle  # noqa: E402
from oracle.vima_oracle import OraclePolicy, ACTION_KEYS  # noqa: E402
from vima_testing import synthetic as syn            # noqa: E402


def ref_dists_to_arrays(dists):
    raw, norm, modes = [], [], []
    for k in ACTION_KEYS:
        d = dists[k]
        norm.append(torch.cat([c.logits for c in d._dists], dim=-1))
        modes.append(d.mode())
    return torch.cat(norm, dim=-1), torch.cat(modes, dim=-1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = sys.argv[1:]
    for name in CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        cfg, wseed, prompts, obs, actions = build_case(name)
        sd = case_state_dict(name, cfg)
        pol = ref_shim.build_reference_policy(**cfg.ctor_kwargs(), xattn_n_positions=cfg.xattn_n_positions)
        missing = pol.load_state_dict(sd, strict=True)   # validates the Appendix-B key/shape contract
        pol.eval()
        obs_ref = {"objects": ref_shim.MapDict(obs["objects"]), "ee": obs["ee"]}
        out, dists = run_policy(pol, prompts, obs_ref, actions)
        # raw (un-normalised) logits: concatenation of the 12 MLP outputs (action_decoder.py:165-166)
        with torch.no_grad():
            raw = torch.cat([torch.cat([m(out["predicted"][-1:]) for m in pol.action_decoder._decoders[k].mlps], dim=-1)
                             for k in ACTION_KEYS], dim=-1)
        norm, modes = ref_dists_to_arrays(dists)
        out["raw_logits"], out["norm_logits"], out["modes"] = raw, norm, modes
        # also de-discretised actions + action tokens of the predicted modes (example.py:195-199)
        acts = {k: dists[k].mode() for k in ACTION_KEYS}
        with torch.no_grad():
            out["mode_action_tokens"] = pol.forward_action_token(acts)
        # oracle cross-check
        orc = OraclePolicy(sd, **cfg.ctor_kwargs())
        o_out, o_d = run_policy(orc, prompts, obs, actions)
        o_out["raw_logits"] = torch.cat([o_d[k]["raw"] for k in ACTION_KEYS], dim=-1)
        o_out["norm_logits"] = torch.cat([torch.cat(o_d[k]["logits"], dim=-1) for k in ACTION_KEYS], dim=-1)
        o_out["modes"] = torch.cat([o_d[k]["mode"] for k in ACTION_KEYS], dim=-1)
        o_out["mode_action_tokens"] = orc.forward_action_token({k: o_d[k]["mode"] for k in ACTION_KEYS})
        print(f"== {name}: cfg={cfg} ({time.time() - t0:.1f}s)")
        for k in out:
            a, b = out[k], o_out[k]
            if a.dtype in (torch.bool, torch.int64):
                print(f"   {k:20s} {tuple(a.shape)} exact={bool((a == b).all())}")
            else:
                diff = (a - b).abs().max().item()
                print(f"   {k:20s} {tuple(a.shape)} max|ref|={a.abs().max().item():.4g} max|oracle-ref|={diff:.3g}")
        arrays = {k: gold_view(name, k, out[k]).detach().cpu().numpy() for k in gold_keys(name, out)}
        arrays["_sd_checksum"] = np.float64(syn.state_dict_checksum(sd))
        arrays["_torch_version"] = np.array(torch.__version__)
        np.savez_compressed(os.path.join(outdir, f"{name}.npz"), **arrays)
        del pol, orc, sd


def main_baselines():
    """tests/golden/baseline_*.npz: outputs of the reference's VIMAGPTPolicy / VIMAGatoPolicy / VIMAFlamingoPolicy."""
    outdir = os.path.join(ROOT, "tests", "golden")
    only = sys.argv[1:]
    for name in BASELINE_CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        cfg, prompts, obs, actions = build_baseline_case(name)
        sd = baseline_state_dict(name, cfg)
        pol = ref_shim.build_reference_baseline(cfg.kind, **cfg.ctor_kwargs())
        pol.load_state_dict(sd, strict=True)
        out = run_baseline(pol, prompts, obs, actions)
        with torch.no_grad():
            out["raw_logits"] = torch.cat([torch.cat([m(out["predicted"][-1:]) for m in pol.action_decoder._decoders[k].mlps], dim=-1)
                                           for k in ACTION_KEYS], dim=-1)
            img = prompts[2]["rgb"]
            out["obj_encoder"] = pol.obj_encoder(rgb=img)
        orc = build_baseline_oracle(cfg, sd)
        o_out = run_baseline(orc, prompts, obs, actions)
        o_out["raw_logits"] = orc.action_logits(o_out["predicted"][-1:])
        o_out["obj_encoder"] = orc.obj_encoder(img)
        print(f"== {name}: cfg={cfg} ({time.time() - t0:.1f}s)")
        for k in out:
            a, b = out[k], o_out[k]
            if a.dtype in (torch.bool, torch.int64):
                print(f"   {k:20s} {tuple(a.shape)} exact={bool((a == b).all())}")
            else:
                print(f"   {k:20s} {tuple(a.shape)} max|ref|={a.abs().max().item():.4g} max|oracle-ref|={(a - b).abs().max().item():.3g}")
        arrays = {k: v.detach().cpu().numpy() for k, v in out.items()}
        arrays["_sd_checksum"] = np.float64(syn.state_dict_checksum(sd))
        arrays["_torch_version"] = np.array(torch.__version__)
        np.savez_compressed(os.path.join(outdir, f"{name}.npz"), **arrays)
        del pol, orc, sd


if __name__ == "__main__":
    main()
    main_baselines()
