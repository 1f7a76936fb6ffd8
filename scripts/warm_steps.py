"""Runs only WARM steps (prompt tokens and K/V cache reused: obs ViT + decoder + action head) or incremental env steps of the
bench workload, for profiling:  python scripts/warm_steps.py [warm|inc] [n] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vima_testing import synthetic as syn  # noqa: E402
from vima_amd.policy import VIMAPolicy  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "warm"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dev = torch.device("cuda", 0)
    cfg = syn.config("200M", xattn_n_positions=512)
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=512, device=dev)
    pol.load_state_dict(syn.make_state_dict(cfg, 0), strict=True)
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        pol.set_option(k, int(v))
    prompts = syn.to_device(syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1236), dev)
    obs = syn.to_device(syn.make_obs(1, B, 4, seed=1336), dev)
    act = syn.to_device(syn.make_actions(1, B, seed=1636), dev)
    ptok, pmask = pol.forward_prompt_assembly(prompts)

    def warm():
        otok, omask = pol.forward_obs_token(obs)
        return pol.action_logits(pol.forward(otok, omask, None, ptok, pmask)[-1])

    def inc(t):
        otok, omask = pol.forward_obs_token(obs)
        atok = pol.forward_action_token(act) if t > 0 else None
        return pol.action_logits(pol.forward_step(otok, omask, atok, ptok, pmask, t))

    if mode == "warm":
        warm(); warm()
    else:
        for t in range(3):
            inc(t)
        inc(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if mode == "warm":
            warm()
        else:
            t = 1 + i % 40                                # steps 1 .. 40 of an episode (n_positions bounds the history), then a new one
            if t == 1 and i > 0:
                inc(0)
            inc(t)
    torch.cuda.synchronize()
    print(f"{mode} batch {B}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")


if __name__ == "__main__":
    main()
