"""vima_amd: MI355X-native (gfx950 / CDNA4) VIMA policy forward pass.

Drop-in for the reference's public entry points on this path:
    vima.create_policy_from_ckpt  (vima/__init__.py:7-16)
    vima.policy.VIMAPolicy        (vima/policy/vima_policy.py:11-322)
    vima.policy.VIMA{GPT,Gato,Flamingo}Policy (vima/policy/vima_*_policy.py; the baselines)
The compute lives in hand-written HIP kernels behind a C ABI (include/vima_hip.h, vima_amd/csrc/);
this package is the thin Python host side. Importing it does not require a GPU; constructing a policy does.
"""
from __future__ import annotations

import os

__all__ = ["VIMAPolicy", "VIMAGPTPolicy", "VIMAGatoPolicy", "VIMAFlamingoPolicy", "create_policy_from_ckpt"]


def __getattr__(name):   # lazy: importing the package must not need the built library
    if name == "VIMAPolicy":
        from .policy import VIMAPolicy
        return VIMAPolicy
    if name in ("VIMAGPTPolicy", "VIMAGatoPolicy", "VIMAFlamingoPolicy"):   # vima/policy/__init__.py:2-4 (baselines)
        from . import baselines
        return getattr(baselines, name)
    raise AttributeError(name)


def create_policy_from_ckpt(ckpt_path, device, precision: str = "bf16"):
    """Same contract as the reference loader (vima/__init__.py:7-16): ckpt = {"cfg": ctor kwargs,
    "state_dict": {"policy.<key>": tensor}}; keys are stripped of the "policy." prefix, loaded strictly, eval()."""
    import torch
    from .policy import VIMAPolicy

    assert os.path.exists(ckpt_path), "Checkpoint path does not exist"
    ckpt = torch.load(ckpt_path, map_location="cpu")
    policy_instance = VIMAPolicy(**ckpt["cfg"], precision=precision, device=device)
    policy_instance.load_state_dict({k.replace("policy.", ""): v for k, v in ckpt["state_dict"].items()}, strict=True)
    policy_instance.eval()
    return policy_instance
