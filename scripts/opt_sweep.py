"""A/B harness on the GPU box: ONE policy / input set (the bench.py headline workload), a list of option combinations,
each timed back to back (warm-up 2, timed N cold steps), repeated R rounds in interleaved order so that clock / thermal
drift hits every variant equally. Usage: python scripts/opt_sweep.py "k=v,k=v" "k=v" ...   ("" = defaults)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vima_testing import synthetic as syn  # noqa: E402
from vima_amd.policy import VIMAPolicy  # noqa: E402

DEFAULTS = {"gemm_tile": 0, "stream_T": 1, "dual_stream": 1, "gemm_persist": 1, "t5_fuse_rms": 1, "vit_chunk": 16384, "gemm_splitk": 0,
            "gemm_small": 1, "attn_qg": 1}


def main():
    combos = sys.argv[1:] or [""]
    steps = int(os.environ.get("STEPS", "6"))
    rounds = int(os.environ.get("ROUNDS", "2"))
    prec = os.environ.get("PREC", "bf16")
    B = int(os.environ.get("BATCH", "256"))
    dev = torch.device("cuda", 0)
    cfg = syn.config("200M", xattn_n_positions=512)
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=512, precision=prec, device=dev)
    pol.load_state_dict(syn.make_state_dict(cfg, 0), strict=True)
    prompts = syn.to_device(syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1236), dev)
    obs = syn.to_device(syn.make_obs(1, B, 4, seed=1336), dev)

    def step():
        ptok, pmask = pol.forward_prompt_assembly(prompts)
        otok, omask = pol.forward_obs_token(obs)
        return pol.action_logits(pol.forward(otok, omask, None, ptok, pmask)[-1])

    res = {c: [] for c in combos}
    ref = None
    for r in range(rounds):
        for c in combos:
            opts = dict(DEFAULTS)
            for kv in filter(None, c.split(",")):
                k, v = kv.split("=")
                opts[k] = int(v)
            for k, v in opts.items():
                pol.set_option(k, v)
            for _ in range(2):
                out = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = step()
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / steps * 1e3)
            if ref is None:
                ref = out.clone()
            d = (out - ref).abs().max().item()
            assert d < float(os.environ.get("TOL", "1e-5")), f"{c}: logits changed by {d}"
    for c in combos:
        print(f"{c or '(defaults)':50s} " + " ".join(f"{x:7.2f}" for x in res[c]) + f"   min {min(res[c]):7.2f} ms")


if __name__ == "__main__":
    main()
