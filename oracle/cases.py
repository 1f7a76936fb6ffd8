"""TEST INFRASTRUCTURE: the parity cases shared by `oracle/make_golden.py` (runs the reference here)
and `tests/` (runs the oracle / the HIP path anywhere). Inputs and weights are regenerated from seeds
(`vima_testing/synthetic.py`); only reference OUTPUTS are stored under tests/golden/."""
from __future__ import annotations

from vima_testing import synthetic as syn

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
    # ---- the BENCHMARKED configurations ------------------------------------------------------------
    # BASELINE.json configs[2] / bench.py default: VIMA-200M, 512-token prompt (32 x [8 words + 1 image -> 8 object
    # tokens]), Q=8, T=1, xattn_n_positions=512: four samples CUT FROM bench.py's own B=256 batch (same seeds 1236/1336)
    "bench_200M": dict(model="200M", npos=512, batch=256, cut=[0, 5, 100, 255], segments=32, words=8, qv=4, steps=1,
                       wseed=0, iseed=1236, obs_seed=1336, strides=dict(prompt_tokens=8)),
    # same inputs, action-head output gain 0.5 instead of the reference's 0.01 -> O(1) logits: argmax agreement of the
    # bf16 path is a meaningful statistic here
    "bench_200M_o1": dict(model="200M", npos=512, batch=256, cut=[1, 2, 3, 64, 128, 129, 200, 254], segments=32, words=8,
                          qv=4, steps=1, wseed=0, iseed=1236, obs_seed=1336, head_gain=0.5,
                          keep=("raw_logits", "norm_logits", "modes", "predicted", "prompt_masks", "obs_masks")),
    # BASELINE.json configs[1]: VIMA-20M, batch 32, 256-token prompt (32 x [4 words + 1 image -> 4 object tokens]), Q=4
    "cfg2_20M": dict(model="20M", npos=256, batch=32, segments=32, words=4, qv=2, steps=1, wseed=0, iseed=1235,
                     strides=dict(prompt_tokens=16)),
    # BASELINE.json configs[4] shape: VIMA-200M, 1024-token prompt (64 x [8 words + 1 image]), xattn_n_positions=1024
    "lp1024_200M": dict(model="200M", npos=1024, batch=2, segments=64, words=8, qv=4, steps=1, wseed=0, iseed=1239,
                        strides=dict(prompt_tokens=16)),
}
# cases cheap enough for every parametrised sweep (the big ones have their own tests)
SMALL_CASES = ("cfg1_T1", "cfg1_T2", "ragged_4M", "e384_long")
BENCH_CASES = ("bench_200M", "bench_200M_o1", "cfg2_20M", "lp1024_200M")


# ---- baseline policies: whole-frame RGB encoders, decoder-only GPT / Perceiver + XAttnGPT ----------
_RAGGED = [[0, 0, 1, 0], [1, 0, 0, 0, 0, 1, 0], [0, 1]]
BASELINE_CASES = {
    "baseline_gpt": dict(kind="gpt", E=256, layers=2, heads=8, batch=3, layout=_RAGGED, steps=3, wseed=3, iseed=2235),
    "baseline_gato": dict(kind="gato", E=256, layers=2, heads=8, batch=3, layout=_RAGGED, steps=2, wseed=4, iseed=2236),
    "baseline_flamingo": dict(kind="flamingo", E=256, layers=2, heads=8, xheads=8, batch=3, layout=_RAGGED, steps=3, wseed=5,
                              iseed=2237),
    # E == 768: no t5_prompt_encoder_post_layer; head dim 64 self-attention; T = 1 without action tokens
    "baseline_gato_768": dict(kind="gato", E=768, layers=1, heads=12, batch=2, segments=2, words=3, steps=1, wseed=6, iseed=2238),
}


def build_baseline_case(name):
    c = BASELINE_CASES[name]
    cfg = syn.BaselineConfig(c["kind"], c["E"], c["layers"], c["heads"], xattn_n_heads=c.get("xheads", 0), vocab_size=16)
    B = c["batch"]
    if "layout" in c:
        prompts = syn.make_rgb_prompt(B, layout=c["layout"], seed=c["iseed"])
    else:
        prompts = syn.make_rgb_prompt(B, n_segments=c["segments"], words_per_segment=c["words"], seed=c["iseed"])
    obs = syn.make_rgb_obs(c["steps"], B, seed=c["iseed"] + 100)
    actions = syn.make_actions(c["steps"] - 1, B, seed=c["iseed"] + 200) if c["steps"] > 1 else None
    return cfg, prompts, obs, actions


def baseline_state_dict(name, cfg=None):
    cfg = cfg or build_baseline_case(name)[0]
    return syn.make_baseline_state_dict(cfg, BASELINE_CASES[name]["wseed"], head_gain=0.5)


def run_baseline(policy, prompts, obs, actions):
    """Drive any object with the method surface of the reference's baseline policies (forward takes no obs mask)."""
    import torch
    with torch.no_grad():
        ptok, pmask = policy.forward_prompt_assembly(prompts)
        otok = policy.forward_obs_token(obs)
        atok = None if actions is None else policy.forward_action_token(actions)
        pred = policy.forward(otok, atok, ptok, pmask)
    out = dict(prompt_tokens=ptok, prompt_masks=pmask, obs_tokens=otok, predicted=pred)
    if atok is not None:
        out["action_tokens"] = atok
    return out


def build_case(name):
    c = CASES[name]
    cfg = syn.config(c["model"], xattn_n_positions=c["npos"])
    B = c["batch"]
    if "layout" in c:
        prompts = syn.make_prompt(B, layout=c["layout"], q_per_view=c["qv"], seed=c["iseed"])
    else:
        prompts = syn.make_prompt(B, n_segments=c["segments"], words_per_segment=c["words"],
                                  q_per_view=c["qv"], seed=c["iseed"])
    obs = syn.make_obs(c["steps"], B, c["qv"], seed=c.get("obs_seed", c["iseed"] + 100))
    actions = syn.make_actions(c["steps"] - 1, B, seed=c["iseed"] + 200) if c["steps"] > 1 else None
    if "cut" in c:
        prompts, obs, actions = syn.cut_prompt(prompts, c["cut"]), syn.cut_obs(obs, c["cut"]), syn.cut_actions(actions, c["cut"])
    return cfg, c["wseed"], prompts, obs, actions


def case_state_dict(name, cfg=None):
    c = CASES[name]
    cfg = cfg or syn.config(c["model"], xattn_n_positions=c["npos"])
    return syn.make_state_dict(cfg, c["wseed"], head_gain=c.get("head_gain", 0.01))


def gold_view(name, key, t):
    """The part of output tensor `key` the fixture of case `name` stores: big cases keep every `stride`-th prompt
    position (dim 0 of the sequence-first [Lp, B, E] tensor) so the committed .npz files stay small."""
    st = CASES[name].get("strides", {}).get(key)
    return t if st is None else t[::st]


def gold_keys(name, keys):
    keep = CASES[name].get("keep")
    return [k for k in keys if keep is None or k in keep]


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
