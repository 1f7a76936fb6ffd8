"""TEST INFRASTRUCTURE: the parity cases shared by `oracle/make_golden.py` (runs the reference here)
and `tests/` (runs the oracle / the HIP path anywhere). Inputs and weights are regenerated from seeds
(`vima_amd/synthetic.py`); only reference OUTPUTS are stored under tests/golden/."""
from __future__ import annotations

from vima_amd import synthetic as syn

# name -> dict(model, xattn_n_positions, batch, layout builder, q_per_view, steps, seeds)
CASES = {
    # BASELINE.json configs[0]: VIMA-2M, batch 1, plumbing case, T=1
    "cfg1_T1": dict(model="2M", npos=256, batch=1, segments=2, words=4, qv=2, steps=1, wseed=0, iseed=1235),
    # same with a two-step history (one past action token)
    "cfg1_T2": dict(model="2M", npos=256, batch=1, segments=2, words=4, qv=2, steps=2, wseed=0, iseed=1236),
    # ragged prompts (padding + masks), 3 steps of history, E=256 N=2
    "ragged_4M": dict(model="4M", npos=256, batch=3, layout=[[0, 0, 1, 0], [1, 0, 0, 0, 0, 1, 0], [0, 1]],
                      qv=3, steps=3, wseed=1, iseed=1237),
    # E != 768 != 256: exercises t5 post layer, 12 heads, xattn_n_positions > 256
    "e384_long": dict(model="20M", npos=320, batch=2, segments=20, words=7, qv=4, steps=2, wseed=2, iseed=1238),
}


def build_case(name):
    c = CASES[name]
    cfg = syn.config(c["model"], xattn_n_positions=c["npos"])
    B = c["batch"]
    if "layout" in c:
        prompts = syn.make_prompt(B, layout=c["layout"], q_per_view=c["qv"], seed=c["iseed"])
    else:
        prompts = syn.make_prompt(B, n_segments=c["segments"], words_per_segment=c["words"],
                                  q_per_view=c["qv"], seed=c["iseed"])
    obs = syn.make_obs(c["steps"], B, c["qv"], seed=c["iseed"] + 100)
    actions = syn.make_actions(c["steps"] - 1, B, seed=c["iseed"] + 200) if c["steps"] > 1 else None
    return cfg, c["wseed"], prompts, obs, actions


def run_policy(policy, prompts, obs, actions):
    """Drive any object exposing the reference VIMAPolicy method surface the way scripts/example.py does
    (prompt once, obs tokens, optional past-action tokens, forward, action decoder). Returns a dict of tensors."""
    import torch
    with torch.no_grad():
        ptok, pmask = policy.forward_prompt_assembly(prompts)
        otok, omask = policy.forward_obs_token(obs)
        atok = None if actions is None else policy.forward_action_token(actions)
        pred = policy.forward(obs_token=otok, obs_mask=omask, action_token=atok,
                              prompt_token=ptok, prompt_token_mask=pmask)
        dists = policy.forward_action_decoder(pred[-1:])
    out = dict(prompt_tokens=ptok, prompt_masks=pmask, obs_tokens=otok, obs_masks=omask, predicted=pred)
    if atok is not None:
        out["action_tokens"] = atok
    return out, dists
