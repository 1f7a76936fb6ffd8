"""The oracle (oracle/vima_oracle.py) against the golden vectors produced by the UNMODIFIED reference
(oracle/make_golden.py, fixtures in tests/golden/). CPU only; this is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle.cases import CASES, build_case, run_policy, case_state_dict, gold_view
from oracle.vima_oracle import OraclePolicy, ACTION_KEYS, t5_relative_position_bucket
from vima_testing import synthetic as syn

# fp32 summation-order noise between the reference's fused modules and the functional restatement
ATOL = 5e-5


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg, wseed, prompts, obs, actions = build_case(name)
    sd = case_state_dict(name, cfg)
    assert abs(syn.state_dict_checksum(sd) - float(gold["_sd_checksum"])) <= 1e-6 * float(gold["_sd_checksum"]), \
        "seeded weights drifted from the ones the fixtures were generated with"
    orc = OraclePolicy(sd, **cfg.ctor_kwargs())
    out, d = run_policy(orc, prompts, obs, actions)
    out["raw_logits"] = torch.cat([d[k]["raw"] for k in ACTION_KEYS], dim=-1)
    out["norm_logits"] = torch.cat([torch.cat(d[k]["logits"], dim=-1) for k in ACTION_KEYS], dim=-1)
    out["modes"] = torch.cat([d[k]["mode"] for k in ACTION_KEYS], dim=-1)
    out["mode_action_tokens"] = orc.forward_action_token({k: d[k]["mode"] for k in ACTION_KEYS})
    for k in gold.files:
        if k.startswith("_"):
            continue
        ref = torch.from_numpy(gold[k])
        got = gold_view(name, k, out[k])
        assert tuple(got.shape) == tuple(ref.shape), k
        if ref.dtype in (torch.bool, torch.int64):
            assert torch.equal(got, ref), k
        else:
            err = (got - ref).abs().max().item()
            assert err <= ATOL, f"{name}/{k}: max abs err {err}"


def test_t5_bucket_known_answers():
    """Known answers of HF's bidirectional bucketing (32 buckets, max distance 128): 8 exact buckets per sign,
    log-spaced to 128, saturating at 15 / 31."""
    rel = torch.tensor([0, -1, -7, -8, -11, -12, -15, -16, -22, -23, -31, -32, -45, -46, -63, -64, -90, -91,
                        -127, -128, -1000, 1, 7, 8, 16, 32, 64, 128, 5000])
    want = torch.tensor([0, 1, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15,
                         15, 15, 15, 17, 23, 24, 26, 28, 30, 31, 31])
    got = t5_relative_position_bucket(rel)
    assert torch.equal(got, want), got


def test_discretize_roundtrip():
    cfg = syn.config("2M")
    orc = OraclePolicy({}, **cfg.ctor_kwargs())
    a = {"pose0_position": torch.tensor([[[0.0, 1.0], [0.5, 0.25]]]),
         "pose0_rotation": torch.tensor([[[0.0, 0.3, 0.6, 1.0], [0.1, 0.2, 0.9, 0.5]]]),
         "pose1_position": torch.tensor([[[0.1, 0.9], [0.99, 0.01]]]),
         "pose1_rotation": torch.tensor([[[0.0, 0.0, 0.0, 1.0], [1.0, 1.0, 1.0, 0.0]]])}
    d = orc.discretize_action(a)
    assert d["pose0_position"].dtype == torch.int64
    assert d["pose0_position"][0, 0].tolist() == [0, 99]
    back = orc._de_discretize_actions(d)
    assert abs(back["pose0_position"][0, 1, 0].item() - 0.5) <= 1 / 50 + 1e-6
