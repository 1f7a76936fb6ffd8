"""Live cross-check of the oracle against the unmodified reference (imported through oracle/ref_shim.py).
Only runs where /root/reference exists (the build container); skipped on the GPU box."""
import pytest
import torch

from oracle import ref_shim
from oracle.cases import build_case, run_policy
from oracle.vima_oracle import OraclePolicy
from vima_testing import synthetic as syn

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def test_live_reference_matches_oracle_masked_prompt():
    """A case that is NOT in the golden set: 9M model (E=320, 10 heads), batch 2, different seeds,
    prompt with a padded tail."""
    cfg = syn.config("9M")
    sd = syn.make_state_dict(cfg, seed=11)
    pol = ref_shim.build_reference_policy(**cfg.ctor_kwargs())
    pol.load_state_dict(sd, strict=True)
    prompts = syn.make_prompt(2, layout=[[0, 1, 0, 0], [0, 0, 1, 0, 1, 0, 0]], q_per_view=2, seed=99)
    obs = syn.make_obs(2, 2, 2, seed=98)
    actions = syn.make_actions(1, 2, seed=97)
    ref_out, ref_d = run_policy(pol, prompts, {"objects": ref_shim.MapDict(obs["objects"]), "ee": obs["ee"]}, actions)
    orc = OraclePolicy(sd, **cfg.ctor_kwargs())
    out, d = run_policy(orc, prompts, obs, actions)
    for k in ref_out:
        if ref_out[k].dtype == torch.bool:
            assert torch.equal(ref_out[k], out[k]), k
        else:
            assert (ref_out[k] - out[k]).abs().max().item() < 5e-5, k
    for k in ref_d:
        assert torch.equal(ref_d[k].mode(), d[k]["mode"]), k
