"""Shared helpers for the -m gpu parity tests (everything goes through the C ABI / the VIMAPolicy host mirror)."""
import ctypes

import torch

from vima_amd import _lib
from vima_testing import synthetic as syn
from vima_amd.policy import VIMAPolicy


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def bf(x):
    return x.bfloat16().float()


def max_rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def max_abs(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


_bare = {}


def bare_policy(prec):
    """Handle without weights: enough for the operator-level entry points."""
    if prec not in _bare:
        p = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8, precision=prec, device="cuda:0")
        p._ensure_handle()
        _bare[prec] = p
    return _bare[prec]


def loaded_policy(cfg, sd, prec, **opts):
    pol = VIMAPolicy(**cfg.ctor_kwargs(), xattn_n_positions=cfg.xattn_n_positions, precision=prec, device="cuda:0")
    pol.load_state_dict(sd, strict=True)
    for k, v in opts.items():
        pol.set_option(k, v)
    return pol
