"""Env-step loop of scripts/example.py (reference lines 135-190) on the MI355X path, with synthetic inputs instead of
the VIMA-Bench simulator (not installable offline): prompt encoded once per episode, then per env step
observation tokens -> decoder -> action distribution -> embedded action for the next step.

    python examples/episode_loop.py [--model 200M] [--batch 1] [--steps 8] [--refeed] [--frames]

--refeed reproduces the reference loop literally (the whole history goes through `forward` every step); the default
uses `forward_step`, which processes only the newest tokens against the episode caches and gives the same predictions.
--frames starts every env step from raw camera frames + segmentation maps (what env.step() returns in the reference) and
runs the GPU image preprocessing (`vima_amd.preprocess.prepare_obs` -> vima_crop_objects: mask -> bbox -> crop -> 32x32
INTER_AREA) instead of feeding ready-made crops; batch 1 like the reference loop.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vima_testing import synthetic as syn                      # noqa: E402
from vima_amd.policy import VIMAPolicy                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="200M")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--refeed", action="store_true")
    ap.add_argument("--frames", action="store_true")
    args = ap.parse_args()
    if args.frames:
        args.batch = 1
    dev = "cuda:0"
    cfg = syn.config(args.model, xattn_n_positions=512)
    policy = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=cfg.xattn_n_positions, precision="bf16", device=dev)
    policy.load_state_dict(syn.make_state_dict(cfg, 0), strict=True)   # create_policy_from_ckpt(path, dev) with a real checkpoint
    B = args.batch
    prompt = syn.to_device(syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1), dev)
    prompt_tokens, prompt_masks = policy.forward_prompt_assembly(prompt)          # once per episode
    if args.frames:     # raw 128 x 256 frames + segmentation of 4 objects per view, as env.step() returns them
        from oracle.preprocess_oracle import synthetic_frames      # (synthetic data generator only)
        observations = []
        for t in range(args.steps):
            fr = {v: synthetic_frames(1, 4, seed=100 + 2 * t + i) for i, v in enumerate(("front", "top"))}
            observations.append({"ee": torch.tensor([t % 2]), "rgb": {v: torch.from_numpy(fr[v][0]).to(dev) for v in fr},
                                 "segm": {v: torch.from_numpy(fr[v][1]).to(dev) for v in fr}, "ids": fr["front"][2]})
    else:
        observations = [syn.to_device(syn.make_obs(1, B, 4, seed=100 + t), dev) for t in range(args.steps)]   # stands in for env.step()
    for episode in range(2):                                                      # episode 0 warms up (workspace, caches)
        ms = run_episode(policy, args, observations, prompt_tokens, prompt_masks)
    print(ms)


def run_episode(policy, args, observations, prompt_tokens, prompt_masks):
    B = args.batch
    obs_cache, act_cache = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        obs = observations[t]
        if args.frames:                                                           # prepare_obs (example.py:374-473) on the GPU
            from vima_amd.preprocess import prepare_obs
            meta = {"n_objects": len(obs["ids"]), "obj_id_to_info": {i: {} for i in obs["ids"]}}
            obs = prepare_obs(obs={"ee": obs["ee"], "rgb": dict(obs["rgb"]), "segm": dict(obs["segm"])}, rgb_dict=None, meta=meta,
                              device=policy._device)
        obs_token, obs_mask = policy.forward_obs_token(obs)                       # [1,B,Q,E], [1,B,Q]
        if args.refeed:
            obs_cache.append((obs_token, obs_mask))
            otok = torch.cat([o for o, _ in obs_cache]), torch.cat([m for _, m in obs_cache])
            atok = torch.cat(act_cache) if act_cache else None
            predicted = policy.forward(otok[0], otok[1], atok, prompt_tokens, prompt_masks)[-1]
        else:
            predicted = policy.forward_step(obs_token, obs_mask, act_cache[-1] if act_cache else None, prompt_tokens,
                                            prompt_masks, step=t)
        dists = policy.forward_action_decoder(predicted.unsqueeze(0))
        actions = {k: v.mode() for k, v in dists.items()}                         # discrete bins, [1,B,n]
        act_cache.append(policy.forward_action_token(actions))                    # [1,B,E] for the next step
        continuous = policy._de_discretize_actions(actions)                       # what env.step() would receive
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    return (f"{args.model} batch {B}: {args.steps} env steps, {ms:.2f} ms per step ({'history re-fed' if args.refeed else 'incremental'}); "
          f"last action pose0_position = {continuous['pose0_position'][0, 0].tolist()}")


if __name__ == "__main__":
    main()
