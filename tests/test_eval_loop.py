"""The reference's evaluation loop through the swap (/root/reference/scripts/example.py:78-240).

`/root/reference` and a GPU never exist on the same machine, so the claim "scripts/example.py runs unchanged on vima_amd" is
proven as a chain:
  1. (CPU, build container) the reference's OWN `main()` -- compiled from its source with `ast`, simulator / tokenizer / cv2
     replaced by synthetic stand-ins (oracle/eval_loop.py) -- and this repository's swap-in loop
     (examples/reference_loop.run_episode) are driven with the SAME recording policy (the unmodified reference VIMAPolicy):
     every policy call, argument, keyword name and result, and every action sent to the environment, must be identical;
  2. (CPU, anywhere) the golden action trace of that run (tests/golden/eval_loop_20M.npz, written by
     `python -m tests.test_eval_loop --write-golden` in the build container) is reproduced by the swap-in loop on the oracle;
  3. (GPU box) the swap-in loop runs on `vima_amd` -- policy AND image preprocessing on the GPU -- and reproduces the golden
     discrete actions of the reference: identical in fp32-operand mode over all env steps, agreement rate reported for bf16;
     the incremental `forward_step` form gives the same actions."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

from oracle import eval_loop, ref_shim                      # noqa: E402
from oracle.vima_oracle import OraclePolicy                 # noqa: E402
from vima_testing import synthetic as syn                       # noqa: E402
import reference_loop                                       # noqa: E402  (examples/)

MODEL, STEPS, N_OBJ, ENV_SEED, WSEED = "20M", 6, 4, 3, 7
MISSING = {2: (3,), 4: (2, 5)}            # objects that vanish in steps 2 / 4: masked slots, reference slot order
GOLDEN = os.path.join(ROOT, "tests", "golden", "eval_loop_20M.npz")
KEYS = ("pose0_position", "pose0_rotation", "pose1_position", "pose1_rotation")


def _env():
    return eval_loop.SyntheticEnv(n_steps=STEPS, n_obj=N_OBJ, seed=ENV_SEED, missing_at=MISSING)


def _state_dict():
    cfg = syn.config(MODEL)
    return cfg, syn.make_state_dict(cfg, WSEED, head_gain=0.5)      # O(1) logits: the argmax is a meaningful statistic


def _run_swap_in(policy, env, device="cpu", incremental=False, gpu_preprocess=False):
    if gpu_preprocess:
        from vima_amd.preprocess import prepare_obs, prepare_prompt_images
    else:
        prepare_obs, prepare_prompt_images = eval_loop.cpu_prepare_obs, eval_loop.cpu_prepare_prompt_images
    with torch.no_grad():
        return reference_loop.run_episode(policy, env, tokenizer=eval_loop.FixedTokenizer(), placeholders=eval_loop.placeholders(),
                                          prepare_obs=prepare_obs, prepare_prompt_images=prepare_prompt_images, device=device,
                                          incremental=incremental)


class _OracleAsPolicy:
    """The oracle behind the policy method surface the loop uses (its action decoder returns plain dicts, the loop wants
    objects with `.mode()` like the reference's MultiCategorical)."""

    class _Mode:
        def __init__(self, m):
            self._m = m

        def mode(self):
            return self._m

    def __init__(self, orc):
        self._orc = orc

    def __getattr__(self, name):
        return getattr(self._orc, name)

    def forward_action_decoder(self, tokens):
        return {k: _OracleAsPolicy._Mode(v["mode"]) for k, v in self._orc.forward_action_decoder(tokens).items()}


def _discrete(records):
    return {k: np.stack([r["discrete"][k].numpy() for r in records]) for k in KEYS}


def _reference_run():
    cfg, sd = _state_dict()
    pol = ref_shim.build_reference_policy(**cfg.ctor_kwargs())
    pol.load_state_dict(sd, strict=True)
    env = _env()
    log = eval_loop.run_reference_loop(pol, env)
    return pol, env, log


@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")
def test_swap_in_loop_makes_exactly_the_reference_loops_calls():
    pol, env_ref, log_ref = _reference_run()
    assert len(env_ref.actions) == STEPS and [c[0] for c in log_ref].count("forward") == STEPS
    rec = eval_loop.RecordingPolicy(pol)
    env = _env()
    records = _run_swap_in(rec, env)
    n = eval_loop.compare_logs(log_ref, rec.log)              # identical tensors at every call boundary
    assert n > 100
    for a, b in zip(env_ref.actions, env.actions):            # what the simulator would have received
        for k in KEYS:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    if os.path.exists(GOLDEN):                                # the committed trace is this run
        gold = np.load(GOLDEN)
        for k in KEYS:
            assert np.array_equal(gold[k], _discrete(records)[k]), k


def test_swap_in_loop_on_the_oracle_reproduces_the_reference_trace():
    """Needs neither the reference nor a GPU: the fp32 oracle (pinned to the reference, tests/test_oracle_golden.py) takes
    the same actions as the reference policy did when the golden trace was recorded."""
    gold = np.load(GOLDEN)
    cfg, sd = _state_dict()
    orc = _OracleAsPolicy(OraclePolicy(sd, **cfg.ctor_kwargs()))
    got = _discrete(_run_swap_in(orc, _env()))
    for k in KEYS:
        assert got[k].shape == gold[k].shape == (STEPS, len(gold[k][0]))
        assert np.array_equal(got[k], gold[k]), (k, got[k], gold[k])
    assert len({tuple(gold["pose0_position"][t]) for t in range(STEPS)}) > 1     # the episode is not a constant action


@pytest.mark.gpu
@pytest.mark.parametrize("incremental", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_reference_eval_loop_on_the_gpu_takes_the_reference_actions(precision, incremental):
    from vima_amd.policy import VIMAPolicy
    gold = np.load(GOLDEN)
    cfg, sd = _state_dict()
    pol = VIMAPolicy(**cfg.ctor_kwargs(), precision=precision, device="cuda:0")
    pol.load_state_dict(sd, strict=True)
    env = _env()
    got = _discrete(_run_swap_in(pol, env, device="cuda:0", incremental=incremental, gpu_preprocess=True))
    total = sum(gold[k].size for k in KEYS)
    agree = sum(int((got[k] == gold[k]).sum()) for k in KEYS)
    first_diff = next((t for t in range(STEPS) if any((got[k][t] != gold[k][t]).any() for k in KEYS)), STEPS)
    print(f"[eval-loop] {precision} {'forward_step' if incremental else 'forward (history re-fed)'}: {agree}/{total} discrete action "
          f"dimensions equal to the reference's over {STEPS} env steps; identical for the first {first_diff} steps")
    assert len(env.actions) == STEPS
    if precision == "fp32":
        assert agree == total          # closed loop: identical actions at every step (>= 5 env steps)
    else:
        assert first_diff >= 1 and agree >= 0.8 * total


def _write_golden():
    _, env, log = _reference_run()
    modes = [c[3] for c in log if c[0] == "forward_action_decoder"]
    out = {k: np.stack([m[k][0, 0].numpy() for m in modes]) for k in KEYS}
    out["continuous_pose0_position"] = np.stack([a["pose0_position"] for a in env.actions])
    os.makedirs(os.path.dirname(GOLDEN), exist_ok=True)
    np.savez(GOLDEN, **out)
    print("wrote", GOLDEN, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if "--write-golden" in sys.argv:
        _write_golden()
