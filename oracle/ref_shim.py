"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (vima_amd/).

Loader that imports the *unmodified* reference (`/root/reference/vima`) inside
this build container so that golden fixtures can be generated from the
reference's own PyTorch modules (see `oracle/make_golden.py`) and so that
`tests/test_oracle_vs_reference.py` can cross-check `oracle/vima_oracle.py`
live.  `/root/reference` does not exist on the GPU box; nothing that runs there
imports this file.

Why shims are needed (each verified in this container):
  * `kornia` / `tree` (dm-tree) are not installed       -> stub modules (need a __spec__)
  * transformers 5.x removed `get_device_map`, `assert_device_map`, `checkpoint`
    from modeling_t5 (only used in dead model-parallel code)  -> dummy attributes
  * `PreTrainedModel.get_head_mask` is gone              -> returns [None]*n
  * `OpenAIGPTModel.get_head_mask` (gpt/gpt.py:173, baselines) likewise
  * HF `Attention.forward` (openai) changed signature; binding the reference's
    `_attn(..., head_mask, output_attentions)` positionally against 5.x would
    silently zero the attention                          -> 4.x parent body restored
  * `from_pretrained("t5-base")` needs the Hub           -> config-initialised random T5-base
No reference source is copied: the shim only patches attributes at import time.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("VIMA_REFERENCE_ROOT", "/root/reference")

T5_BASE = dict(
    vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12,
    relative_attention_num_buckets=32, relative_attention_max_distance=128,
    dropout_rate=0.1, layer_norm_epsilon=1e-6, feed_forward_proj="relu",
    is_encoder_decoder=False, use_cache=False,
)

_loaded = None


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vima"))


def _stub(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def load_reference():
    """Import the reference package with the compatibility shims applied. Returns the `vima` module."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    if "kornia" not in sys.modules:
        _stub("kornia")
    if "tree" not in sys.modules:
        tree = _stub("tree")

        def map_structure(func, *structs):
            """Minimal dm-tree `map_structure` (dict / list / tuple nests) for vima/utils.py any_* helpers."""
            s0 = structs[0]
            import collections.abc as cabc
            if isinstance(s0, cabc.Mapping):
                return type(s0)({k: map_structure(func, *[s[k] for s in structs]) for k in s0.keys()})
            if isinstance(s0, (list, tuple)):
                return type(s0)(map_structure(func, *xs) for xs in zip(*structs))
            return func(*structs)

        tree.map_structure = map_structure

        def traverse(fn, structure, top_down=True):
            """Minimal dm-tree `traverse`: visit every sub-structure; a non-None return value replaces it (bottom-up:
            children first). Only what vima/utils.py::_wrap_datadict needs."""
            import collections.abc as cabc
            if top_down:
                r = fn(structure)
                if r is not None:
                    return r
            if isinstance(structure, cabc.Mapping):
                structure = type(structure)({k: traverse(fn, v, top_down) for k, v in structure.items()})
            elif isinstance(structure, (list, tuple)):
                structure = type(structure)(traverse(fn, v, top_down) for v in structure)
            if not top_down:
                r = fn(structure)
                if r is not None:
                    return r
            return structure

        tree.traverse = traverse

        def _is_map(x):
            import collections.abc as cabc
            return isinstance(x, cabc.Mapping)

        def map_structure_with_path(func, *structs, _path=()):
            s0 = structs[0]
            if _is_map(s0):
                return type(s0)({k: map_structure_with_path(func, *[s[k] for s in structs], _path=_path + (k,)) for k in s0.keys()})
            if isinstance(s0, (list, tuple)):
                return type(s0)(map_structure_with_path(func, *xs, _path=_path + (i,)) for i, xs in enumerate(zip(*structs)))
            return func(_path, *structs)

        def flatten(structure):
            """dm-tree order: mappings by sorted key, sequences in order."""
            if _is_map(structure):
                return [leaf for k in sorted(structure.keys()) for leaf in flatten(structure[k])]
            if isinstance(structure, (list, tuple)):
                return [leaf for v in structure for leaf in flatten(v)]
            return [structure]

        def unflatten_as(structure, flat):
            it = iter(flat)

            def rec(s):
                if _is_map(s):
                    vals = {k: rec(s[k]) for k in sorted(s.keys())}
                    return type(s)({k: vals[k] for k in s.keys()})
                if isinstance(s, (list, tuple)):
                    return type(s)(rec(v) for v in s)
                return next(it)
            return rec(structure)

        tree.map_structure_with_path = map_structure_with_path
        tree.flatten = flatten
        tree.unflatten_as = unflatten_as
    import transformers.models.t5.modeling_t5 as mt5
    from transformers import T5Config

    for n in ("get_device_map", "assert_device_map", "checkpoint"):
        if not hasattr(mt5, n):
            setattr(mt5, n, None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import vima  # noqa: F401

    PE = importlib.import_module("vima.nn.prompt_encoder.prompt_encoder")
    if not isinstance(PE, types.ModuleType) or not hasattr(PE, "T5Stack"):
        PE = sys.modules["vima.nn.prompt_encoder.prompt_encoder"]
    WE = sys.modules["vima.nn.prompt_encoder.word_embd"]
    C = sys.modules["vima.nn.seq_modeling.xattn_gpt.components"]

    PE.T5Stack.get_head_mask = lambda self, hm, n, *a, **k: [None] * n

    def _t5_from_pretrained(cls, name, *a, **k):
        return cls(T5Config(**T5_BASE))

    PE.T5EncoderModel.from_pretrained = classmethod(_t5_from_pretrained)

    class _FakeAutoModel:
        @staticmethod
        def from_pretrained(name, *a, **k):
            class _M:
                def get_input_embeddings(self):
                    return nn.Embedding(T5_BASE["vocab_size"], T5_BASE["d_model"])
            return _M()

    WE.AutoModel = _FakeAutoModel

    def attn_fwd(self, x, attention_mask=None, head_mask=None, output_attentions=False):
        # transformers 4.x `modeling_openai.Attention.forward` body
        x = self.c_attn(x)
        q, k, v = x.split(self.split_size, dim=2)
        q = self.split_heads(q)
        k = self.split_heads(k, k=True)
        v = self.split_heads(v)
        o = self._attn(q, k, v, attention_mask, head_mask, output_attentions)
        a = self.merge_heads(o[0])
        a = self.c_proj(a)
        a = self.resid_dropout(a)
        return [a] + o[1:]

    C.Attention.forward = attn_fwd
    # the decoder-only baselines (VIMAGPTPolicy / VIMAGatoPolicy) use a second copy of the same classes (gpt/gpt.py)
    G = sys.modules["vima.nn.seq_modeling.gpt.gpt"]
    G.Attention.forward = attn_fwd
    G.OpenAIGPTModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    _loaded = vima
    return vima


class MapDict(dict):
    """Plain nested dict with the `.map_structure(func=)` hook `VIMAPolicy.forward_obs_token` needs
    (reference: vima/policy/vima_policy.py:246; the real type is vima.utils.DataDict)."""

    def map_structure(self, func):
        def rec(x):
            if isinstance(x, dict):
                return MapDict({k: rec(v) for k, v in x.items()})
            return func(x)
        return rec(self)


def build_reference_policy(embed_dim, xf_n_layers, sattn_n_heads, xattn_n_heads, xattn_n_positions=256):
    """Construct the reference VIMAPolicy; swaps in a larger-`xattn_n_positions` XAttnGPT when the
    prompt exceeds the reference's hard-coded 256."""
    vima = load_reference()
    import vima.nn as vnn
    pol = vima.policy.VIMAPolicy(embed_dim=embed_dim, xf_n_layers=xf_n_layers,
                                 sattn_n_heads=sattn_n_heads, xattn_n_heads=xattn_n_heads)
    if xattn_n_positions != 256:
        pol.xattn_gpt = vnn.XAttnGPT(
            embed_dim, n_layer=xf_n_layers, n_head=sattn_n_heads, dropout=0.1,
            xattn_n_head=xattn_n_heads, xattn_ff_expanding=4,
            xattn_n_positions=xattn_n_positions, use_geglu=True)
    pol.eval()
    return pol


def build_reference_baseline(kind, **ctor):
    """Construct one of the reference's baseline policies (vima/policy/vima_{gpt,gato,flamingo}_policy.py) in eval mode.
    They read `self.device`, which plain nn.Module does not define (the reference's training harness does): set it."""
    vima = load_reference()
    cls = {"gpt": vima.policy.VIMAGPTPolicy, "gato": vima.policy.VIMAGatoPolicy, "flamingo": vima.policy.VIMAFlamingoPolicy}[kind]
    pol = cls(**ctor).eval()
    pol.device = torch.device("cpu")
    return pol
