"""Build-container only (needs /root/reference): times the UNMODIFIED reference (through oracle/ref_shim.py) and the oracle
port on the same host cores on the bench workload at a small batch, so that bench.py's `cpu_baseline` (kind "port", the
only thing that can run on the GPU box) can be related to the reference itself. Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import ref_shim  # noqa: E402
from oracle.cases import run_policy  # noqa: E402
from oracle.vima_oracle import OraclePolicy  # noqa: E402
from vima_testing import synthetic as syn  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(os.cpu_count())
    cfg = syn.config("200M", xattn_n_positions=512)
    sd = syn.make_state_dict(cfg, 0)
    prompts = syn.make_prompt(B, n_segments=32, words_per_segment=8, q_per_view=4, seed=1236)
    obs = syn.make_obs(1, B, 4, seed=1336)
    ref = ref_shim.build_reference_policy(**cfg.ctor_kwargs(), xattn_n_positions=512)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    orc = OraclePolicy(sd, **cfg.ctor_kwargs())
    obs_ref = {"objects": ref_shim.MapDict(obs["objects"]), "ee": obs["ee"]}

    def timed(fn, n=3):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        t_ref = timed(lambda: run_policy(ref, prompts, obs_ref, None))
        t_orc = timed(lambda: run_policy(orc, prompts, obs, None))
    print(json.dumps({"workload": f"VIMA-200M cold step, Lp=512, Q=8, batch {B}, fp32", "threads": torch.get_num_threads(),
                      "host_cpu_count": os.cpu_count(), "reference_s_per_step": round(t_ref, 3), "oracle_port_s_per_step": round(t_orc, 3),
                      "reference_samples_per_s": round(B / t_ref, 3), "oracle_port_samples_per_s": round(B / t_orc, 3),
                      "port_over_reference_speed": round(t_ref / t_orc, 3)}))


if __name__ == "__main__":
    main()
