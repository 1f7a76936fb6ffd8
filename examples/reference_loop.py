"""The reference's evaluation loop (`main()`, /root/reference/scripts/example.py:97-240) on the `vima_amd` method surface,
as a function: ONE episode of

    prompt -> forward_prompt_assembly (once) ; per env step: frames -> prepare_obs -> forward_obs_token -> forward over the
    whole history (or forward_step, the incremental form) -> forward_action_decoder -> mode -> forward_action_token (next
    step's history) -> _de_discretize_actions -> scale to the task's action bounds -> env.step

with the environment, the tokenizer and the two image-preprocessing functions passed in. With `vima_amd.preprocess.prepare_obs`
/ `prepare_prompt_images` and a `vima_amd` policy everything from the camera frames on runs on the GPU; the call sequence --
every policy method, its arguments and keyword names -- is the reference loop's: tests/test_eval_loop.py runs the
reference's own `main()` (compiled from its source with `ast`) and this function against the same recording policy and
requires identical call logs, then the GPU test runs this function on the box where /root/reference does not exist.

    python examples/reference_loop.py [--model 20M] [--steps 6] [--incremental]     (synthetic environment, random weights)
"""
from __future__ import annotations

import numpy as np
import torch

VIEWS = ("front", "top")


def encode_prompt(prompt, prompt_assets, *, tokenizer, placeholders, prepare_prompt_images, device, views=VIEWS):
    """The `prompts` triple of `forward_prompt_assembly` from a prompt string and its assets (example.py:243-371): token types
    (0 word / 1 image) of the single prompt, the word ids, and the padded object crops of every placeholder in prompt order."""
    enc = tokenizer.encode(prompt, add_special_tokens=True)
    names = [tok[1:-1] for tok in enc.tokens if tok in placeholders]
    assert set(prompt_assets.keys()) == set(names)
    token_type = [1 if tok in placeholders else 0 for tok in enc.tokens]
    words = [i for i, tok in zip(enc.ids, enc.tokens) if tok not in placeholders]
    word_batch = torch.tensor(words, dtype=torch.int64, device=device)
    image_batch = prepare_prompt_images(prompt_assets, names, views=views, device=device)
    return [token_type], word_batch, image_batch


def run_episode(policy, env, *, tokenizer, placeholders, prepare_obs, prepare_prompt_images, device, views=VIEWS, incremental=False):
    """One episode; returns the per-step records {"discrete": {key: int64 [n]}, "continuous": {key: float32 [n]}} (what
    `env.step` received is `continuous`)."""
    device = torch.device(device)
    obs = env.reset()
    env.render()
    meta = env.meta_info
    prompt_tokens, prompt_masks = policy.forward_prompt_assembly(
        encode_prompt(env.prompt, env.prompt_assets, tokenizer=tokenizer, placeholders=placeholders,
                      prepare_prompt_images=prepare_prompt_images, device=device, views=views))
    low = torch.tensor(np.asarray([meta["action_bounds"]["low"]]), dtype=torch.float32, device=device)
    high = torch.tensor(np.asarray([meta["action_bounds"]["high"]]), dtype=torch.float32, device=device)
    hist_tok, hist_mask, hist_act = [], [], []          # per step: [Q, E], [Q], [E]
    records = []
    step = 0
    while True:
        frame = {"ee": np.asarray(obs["ee"])[None],                                     # a history axis of length 1
                 "rgb": {v: np.asarray(obs["rgb"][v])[None] for v in views},
                 "segm": {v: np.asarray(obs["segm"][v])[None] for v in views}}
        tok, msk = policy.forward_obs_token(prepare_obs(obs=frame, rgb_dict=None, meta=meta, device=device))   # [1, 1, Q, E]
        if incremental:
            prev = hist_act[-1][None, None] if hist_act else None
            predicted = policy.forward_step(tok, msk, prev, prompt_tokens, prompt_masks, step=step)            # [1, E]
            predicted = predicted.unsqueeze(0)
        else:
            hist_tok.append(tok[0, 0])
            hist_mask.append(msk[0, 0])
            q = max(t.shape[0] for t in hist_tok)                                       # pad every step to the most objects
            toks = torch.stack([torch.cat([t, t.new_zeros(q - t.shape[0], t.shape[1])]) for t in hist_tok]).unsqueeze(1)
            msks = torch.stack([torch.cat([m, m.new_zeros(q - m.shape[0])]) for m in hist_mask]).unsqueeze(1)
            acts = torch.stack(hist_act).unsqueeze(1) if hist_act else None             # [T - 1, 1, E]
            predicted = policy.forward(obs_token=toks, action_token=acts, prompt_token=prompt_tokens,
                                       prompt_token_mask=prompt_masks, obs_mask=msks)[-1].unsqueeze(0)         # [1, 1, E]
        dists = policy.forward_action_decoder(predicted)
        discrete = {k: d.mode() for k, d in dists.items()}
        hist_act.append(policy.forward_action_token(discrete)[0, 0])
        cont = policy._de_discretize_actions(discrete)
        for k in ("pose0_position", "pose1_position"):
            cont[k] = torch.clamp(cont[k] * (high - low) + low, min=low, max=high)
        for k in ("pose0_rotation", "pose1_rotation"):
            cont[k] = torch.clamp(cont[k] * 2 - 1, min=-1, max=1)
        to_env = {k: v.cpu().numpy()[0, 0] for k, v in cont.items()}
        records.append({"discrete": {k: v[0, 0].cpu() for k, v in discrete.items()}, "continuous": to_env})
        obs, _, done, _ = env.step(to_env)
        step += 1
        if done:
            return records


def main():
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.eval_loop import FixedTokenizer, SyntheticEnv, placeholders            # synthetic stand-ins for VIMA-Bench / the Hub
    from vima_testing import synthetic as syn
    from vima_amd.policy import VIMAPolicy
    from vima_amd.preprocess import prepare_obs, prepare_prompt_images
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="20M")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--incremental", action="store_true")
    args = ap.parse_args()
    cfg = syn.config(args.model)
    policy = VIMAPolicy(**cfg.ctor_kwargs(), precision="bf16", device="cuda:0")
    policy.load_state_dict(syn.make_state_dict(cfg, 0, head_gain=0.5), strict=True)
    recs = run_episode(policy, SyntheticEnv(n_steps=args.steps), tokenizer=FixedTokenizer(), placeholders=placeholders(),
                       prepare_obs=prepare_obs, prepare_prompt_images=prepare_prompt_images, device="cuda:0", incremental=args.incremental)
    for t, r in enumerate(recs):
        print(t, {k: v.tolist() for k, v in r["discrete"].items()})


if __name__ == "__main__":
    main()
