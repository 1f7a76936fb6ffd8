"""`VIMAPolicy`: drop-in host-side mirror of the reference policy class, backed by the gfx950 HIP library.

Mirrors the method surface, argument meaning, return shapes and error behaviour of
`vima/policy/vima_policy.py:11-322` so that `scripts/example.py` (the reference's only caller) runs unchanged:

    forward / __call__            vima_policy.py:116-159   -> vima_decode
    forward_prompt_assembly       vima_policy.py:161-240   -> vima_prompt_encode
    forward_obs_token             vima_policy.py:242-259   -> vima_obs_encode
    forward_action_token          vima_policy.py:261-262   -> vima_action_embed
    forward_action_decoder        vima_policy.py:264-265   -> vima_action_head (+ MultiCategorical wrapper)
    discretize_action             vima_policy.py:267-299
    _de_discretize_actions        vima_policy.py:301-322
    load_state_dict(strict=True)  vima/__init__.py:11-14   -> vima_set_param / vima_finalize_params

PyTorch is only the tensor container / allocator / stream provider here; all arithmetic of the hot path runs in the
hand-written HIP kernels. There is no CPU fallback: constructing a policy without a GPU raises.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .dists import MultiCategorical

VIEWS = ("front", "top")  # sorted(["front", "top"]) (obj_encoder.py:31, vima_policy.py:110)
ACTION_KEYS = ("pose0_position", "pose0_rotation", "pose1_position", "pose1_rotation")
ACTION_DIMS = OrderedDict([("pose0_position", [50, 100]), ("pose0_rotation", [50] * 4),
                           ("pose1_position", [50, 100]), ("pose1_rotation", [50] * 4)])  # vima_policy.py:82-87
N_LOGITS = 700


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _pair(a, b):
    return (ctypes.c_void_p * 2)(a, b)


def build_prompt_index(raw_types, Q: int):
    """Index form of the prompt assembly loops (vima_policy.py:168-233), vectorised per sample: for every sample the
    output positions of its tokens in order, a word taking one slot and an image Q slots (front objects, then top
    objects). Returns (src int32 [B, L_max], L_max, n_words, n_images) with src >= 0: index into word_batch; -1: padding;
    <= -2: object slot -(image_index * Q + j) - 2. Raises ValueError for token types other than 0 / 1 like the reference
    (vima_policy.py:177)."""
    rows, L_max, wp, ip = [], 0, 0, 0
    for p in raw_types:
        t = np.asarray(p, dtype=np.int64).reshape(-1)
        bad = t[(t != 0) & (t != 1)]
        if bad.size:
            raise ValueError(f"Invalid prompt token type {int(bad[0])}")
        is_img = t == 1
        lens = np.where(is_img, Q, 1)
        starts = np.cumsum(lens) - lens
        L_this = int(lens.sum())
        row = np.empty(L_this, dtype=np.int32)
        nw, ni = int((~is_img).sum()), int(is_img.sum())
        row[starts[~is_img]] = wp + np.arange(nw, dtype=np.int32)
        if ni and Q:
            pos = (starts[is_img][:, None] + np.arange(Q)[None, :]).reshape(-1)
            row[pos] = -(((ip + np.arange(ni))[:, None] * Q + np.arange(Q)[None, :]).reshape(-1) + 2)
        wp += nw
        ip += ni
        rows.append(row)
        L_max = max(L_max, L_this)
    src = np.full((len(rows), max(L_max, 1)), -1, dtype=np.int32)
    for b, row in enumerate(rows):
        src[b, :row.shape[0]] = row
    return src, L_max, wp, ip


class VIMAPolicy(nn.Module):
    def __init__(self, *, embed_dim: int, xf_n_layers: int, sattn_n_heads: int, xattn_n_heads: int,
                 xattn_n_positions: int = 256, n_positions: int = 512, precision: str = "bf16", device=None):
        super().__init__()
        if embed_dim % xattn_n_heads != 0:   # components.py:120-123
            raise ValueError(f"dim ({embed_dim}) must be divisible by num_heads ({xattn_n_heads}).")
        if embed_dim % sattn_n_heads != 0:   # HF Attention.__init__
            raise ValueError(f"Attention n_state shape: {embed_dim} must be divisible by config.n_head {sattn_n_heads}")
        if precision not in _lib.PRECISION:
            raise ValueError(f"precision must be one of {list(_lib.PRECISION)}")
        self.embed_dim = embed_dim
        self.precision = precision
        self._cfg = _lib.VimaConfig(embed_dim, xf_n_layers, sattn_n_heads, xattn_n_heads, xattn_n_positions,
                                    n_positions, _lib.PRECISION[precision])
        self._cfg_kwargs = dict(embed_dim=embed_dim, xf_n_layers=xf_n_layers, sattn_n_heads=sattn_n_heads,
                                xattn_n_heads=xattn_n_heads)
        self._lib = _lib.load()
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        self._handle = None
        self._sd = None
        self._views = list(VIEWS)
        self._n_discrete_x_bins = 50
        self._n_discrete_y_bins = 100
        self._n_discrete_z_bins = 50
        self._n_discrete_rot_bins = 50
        self._input_checked = False
        self._img_checked = False
        # cross-step prompt K/V cache: the eval loop passes the SAME prompt tensor every env step
        # (scripts/example.py:118,184-190); keyed on storage pointer + in-place version counter + layout
        self.cache_prompt_kv = True
        self._kv_key = None

    # ------------------------------------------------------------------ lifetime / weights
    def _ensure_handle(self):
        if self._handle is not None:
            return
        if self._device is None or self._device.type != "cuda":
            raise RuntimeError("VIMAPolicy (vima_amd) needs an AMD GPU device: the HIP library has no CPU fallback")
        h = ctypes.c_void_p()
        idx = self._device.index if self._device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.vima_create(ctypes.byref(self._cfg), idx, ctypes.byref(h)))
        self._handle = h

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                self._lib.vima_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def expected_keys(self):
        """(required, ignorable) reference state_dict keys for this config."""
        req = _lib.required_params(self._cfg)
        n = self._cfg.xf_n_layers
        ign = ["xattn_gpt.position_ids", "xattn_gpt.xattn_position_ids",
               "t5_prompt_encoder.t5.shared.weight", "t5_prompt_encoder.t5.encoder.embed_tokens.weight"]
        ign += [f"xattn_gpt.h.{i}.attn.bias" for i in range(n)]
        ign += [f"xattn_gpt.xattns.{i}.kv_position_ids" for i in range(n)]
        return req, ign

    def load_state_dict(self, state_dict, strict: bool = True):
        req, ign = self.expected_keys()
        given = set(state_dict.keys())
        missing = [k for k in req if k not in given]
        unexpected = sorted(given - set(req) - set(ign))
        if strict:
            missing_ign = [k for k in ign if k not in given]
            if missing or unexpected or missing_ign:
                raise RuntimeError(
                    f"Error(s) in loading state_dict for {type(self).__name__}:\n"
                    f"\tMissing key(s) in state_dict: {missing + missing_ign}.\n"
                    f"\tUnexpected key(s) in state_dict: {unexpected}.")
        elif missing:
            raise RuntimeError(f"vima_amd cannot run with missing weights: {missing}")
        self._kv_key = None
        if self._handle is not None:   # re-loading: start from a fresh handle
            self._lib.vima_destroy(self._handle)
            self._handle = None
        self._ensure_handle()
        for k in req:
            t = state_dict[k].detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check(self._lib.vima_set_param(self._handle, k.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))
        _lib.check(self._lib.vima_finalize_params(self._handle))
        self._sd = OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def state_dict(self, *args, **kwargs):
        return OrderedDict() if self._sd is None else OrderedDict(self._sd)

    def to(self, device=None, *args, **kwargs):
        if device is not None and not isinstance(device, torch.dtype):
            device = torch.device(device)
            if device.type != "cuda":
                raise RuntimeError("vima_amd.VIMAPolicy only runs on an AMD GPU (device 'cuda[:i]'); no CPU fallback")
            if self._handle is not None and device != self._device:
                sd = self._sd
                self._lib.vima_destroy(self._handle)
                self._handle = None
                self._device = device
                if sd is not None:
                    self.load_state_dict(sd, strict=True)
            self._device = device
        return self

    def set_option(self, key: str, value: int):
        self._ensure_handle()
        _lib.check(self._lib.vima_set_option(self._handle, key.encode(), int(value)))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _ready(self):
        if self._handle is None or self._sd is None:
            raise RuntimeError("VIMAPolicy weights not loaded: call load_state_dict() first")

    # ------------------------------------------------------------------ object / observation tokens
    def _views_of(self, d, what):
        out = []
        for v in VIEWS:
            if v not in d:
                raise KeyError(f"{what} is missing view '{v}'")
            out.append(d[v])
        return out

    def _check_img(self, crops):
        if not self._img_checked:   # preprocess.py:20-28 (first call only: it is a device->host sync)
            for c in crops:
                assert torch.is_tensor(c) and c.dim() >= 4
                if c.numel():
                    assert c.max() > 2, "img should be between [0, 255] before normalize"
            self._img_checked = True

    def _obj_inputs(self, objects, lead_dims):
        crops = [c.reshape(-1, *c.shape[lead_dims:]) for c in self._views_of(objects["cropped_img"], "cropped_img")]
        bbox = [b.reshape(-1, *b.shape[lead_dims:]) for b in self._views_of(objects["bbox"], "bbox")]
        mask = [m.reshape(-1, m.shape[-1]) for m in self._views_of(objects["mask"], "mask")]
        n, qv = crops[0].shape[0], crops[0].shape[1]
        for c, b, m in zip(crops, bbox, mask):
            if tuple(c.shape) != (n, qv, 3, 32, 32):
                raise ValueError(f"cropped_img must be [..., {qv}, 3, 32, 32] for every view, got {tuple(c.shape)}")
            if tuple(b.shape) != (n, qv, 4) or tuple(m.shape) != (n, qv):
                raise ValueError("bbox / mask shapes do not match cropped_img")
        self._check_img(crops)
        dev = self._device
        crops = [c.to(device=dev, dtype=torch.uint8).contiguous() for c in crops]
        bbox = [b.to(device=dev, dtype=torch.int64).contiguous() for b in bbox]
        mask = [m.to(device=dev, dtype=torch.bool).contiguous() for m in mask]
        return crops, bbox, mask, n, qv

    def forward_obs_token(self, obs):
        """obs = {"objects": {cropped_img,bbox,mask}{view} with leading [L_obs, B], "ee": [L_obs, B] int64}
        -> (obs_feats [L_obs, B, Q, E] fp32, obj_mask [L_obs, B, Q] bool)   (vima_policy.py:242-259)"""
        self._ready()
        objects, ee = obs["objects"], obs["ee"]
        lead = tuple(ee.shape[:2])
        crops, bbox, mask, n, qv = self._obj_inputs(objects, 2)
        assert n == lead[0] * lead[1]
        ee = ee.to(device=self._device, dtype=torch.int64).contiguous()
        E = self.embed_dim
        out = torch.empty(n, 2 * qv, E, dtype=torch.float32, device=self._device)
        omask = torch.empty(n, 2 * qv, dtype=torch.bool, device=self._device)
        _lib.check(self._lib.vima_obs_encode(
            self._handle, _pair(crops[0].data_ptr(), crops[1].data_ptr()), _pair(bbox[0].data_ptr(), bbox[1].data_ptr()),
            _pair(mask[0].data_ptr(), mask[1].data_ptr()), _ptr(ee), n, qv, _ptr(out), _ptr(omask), self._stream()))
        return out.view(*lead, 2 * qv, E), omask.view(*lead, 2 * qv)

    def t5_encode(self, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """T5PromptEncoder.forward on already assembled embeddings (prompt_encoder.py:30-58): x [B,L,768] fp32,
        mask [B,L] bool -> [B,L,768] (before t5_prompt_encoder_post_layer). Sub-module level entry for parity tests."""
        self._ready()
        dev = self._device
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        mask = mask.to(device=dev, dtype=torch.bool).contiguous()
        B, L, _ = x.shape
        out = torch.empty_like(x)
        _lib.check(self._lib.vima_t5_encode(self._handle, _ptr(x), _ptr(mask), B, L, _ptr(out), self._stream()))
        return out

    def obj_encoder(self, cropped_img, bbox, mask=None):
        """ObjEncoder.forward (obj_encoder.py:66-95) for inputs with ONE leading dim: -> [n, 2*Qv, E]."""
        self._ready()
        objects = {"cropped_img": cropped_img, "bbox": bbox,
                   "mask": mask if mask is not None else {v: torch.ones(bbox[v].shape[:-1], dtype=torch.bool) for v in VIEWS}}
        crops, bb, _, n, qv = self._obj_inputs(objects, 1)
        out = torch.empty(n, 2 * qv, self.embed_dim, dtype=torch.float32, device=self._device)
        _lib.check(self._lib.vima_obj_encode(self._handle, _pair(crops[0].data_ptr(), crops[1].data_ptr()),
                                             _pair(bb[0].data_ptr(), bb[1].data_ptr()), n, qv, _ptr(out), self._stream()))
        return out

    # ------------------------------------------------------------------ prompt
    def forward_prompt_assembly(self, prompts):
        """prompts = (raw_prompts_token_type: list[list[0|1]], word_batch [n_words] int64, image_batch)
        -> (prompt_tokens [L, B, E] fp32, prompt_masks [B, L] bool)   (vima_policy.py:161-240)"""
        self._ready()
        raw_types, word_batch, image_batch = prompts
        n_img_total = sum(1 for p in raw_types for t in p if t == 1)
        dev = self._device
        if n_img_total > 0:
            crops, bbox, mask, n_img, qv = self._obj_inputs(image_batch, 1)
        else:
            crops = bbox = mask = None
            n_img, qv = 0, 0
            try:
                qv = int(image_batch["cropped_img"][VIEWS[0]].shape[1])
            except Exception:
                qv = 0
        Q = 2 * qv
        B = len(raw_types)
        src, L_max, wp, ip = build_prompt_index(raw_types, Q)
        if wp > word_batch.numel() or ip > n_img:
            raise IndexError("prompt token types reference more words / images than provided")
        word_batch = word_batch.to(device=dev, dtype=torch.int64).contiguous()
        tok_src = torch.from_numpy(src[:, :L_max].copy()).to(dev)
        out = torch.empty(B, L_max, self.embed_dim, dtype=torch.float32, device=dev)
        omask = torch.empty(B, L_max, dtype=torch.bool, device=dev)
        z = 0
        _lib.check(self._lib.vima_prompt_encode(
            self._handle, _ptr(word_batch), int(word_batch.numel()),
            _pair(crops[0].data_ptr(), crops[1].data_ptr()) if crops else _pair(z, z),
            _pair(bbox[0].data_ptr(), bbox[1].data_ptr()) if bbox else _pair(z, z),
            _pair(mask[0].data_ptr(), mask[1].data_ptr()) if mask else _pair(z, z),
            n_img, qv, _ptr(tok_src), B, L_max, _ptr(out), _ptr(omask), self._stream()))
        return out.transpose(0, 1), omask

    # ------------------------------------------------------------------ decoder
    def _check_input(self, obs_token, prompt_token, prompt_mask, obs_mask):
        """XAttnGPT._check_input (xattn_gpt.py:141-177), first call only like the reference."""
        assert prompt_token.dim() == 3
        assert prompt_token.dtype == torch.float32
        assert obs_token.dtype == torch.float32
        L_p, B_p, E_p = prompt_token.shape
        assert B_p == obs_token.shape[1]
        assert E_p == obs_token.shape[-1]
        if prompt_mask is not None:
            assert prompt_mask.shape == (B_p, L_p), \
                f"Expect `prompt_mask` to have shape of ({B_p, L_p}), but got {tuple(prompt_mask.shape)}"
            assert torch.all(prompt_mask.sum(dim=-1) > 0), "each source token should attend to at least one target token"
            assert prompt_mask.dtype == torch.bool
        assert obs_mask.dtype == torch.bool
        assert torch.all(obs_mask[0, :, 0]), "first observation token of every sample must be valid (position id >= 0)"

    def forward(self, obs_token: torch.Tensor, obs_mask: torch.Tensor, action_token: torch.Tensor | None,
                prompt_token: torch.Tensor, prompt_token_mask: torch.Tensor):
        """-> predicted_action_tokens [L_obs, B, E] fp32, sequence-first (vima_policy.py:116-159)."""
        self._ready()
        L_obs, B, Q, E = obs_token.shape
        if not self._input_checked:
            self._check_input(obs_token, prompt_token, prompt_token_mask, obs_mask)
            self._input_checked = True
        dev = self._device
        # cheap host-side checks on EVERY call (the reference asserts fp32 in _check_input; the raw-pointer boundary would
        # silently reinterpret any other dtype): shapes + dtypes, no device synchronisation
        if obs_token.dtype != torch.float32 or prompt_token.dtype != torch.float32:
            raise AssertionError(f"obs_token / prompt_token must be float32 (xattn_gpt.py:150,152), got {obs_token.dtype} / {prompt_token.dtype}")
        if prompt_token.dim() != 3 or prompt_token.shape[1] != B or prompt_token.shape[2] != E:
            raise AssertionError(f"prompt_token must be [Lp, {B}, {E}], got {tuple(prompt_token.shape)}")
        if tuple(obs_mask.shape) != (L_obs, B, Q) or tuple(prompt_token_mask.shape) != (B, prompt_token.shape[0]):
            raise AssertionError("obs_mask / prompt_token_mask shapes do not match the tokens")
        obs_token = obs_token.to(device=dev, dtype=torch.float32).contiguous()
        obs_mask = obs_mask.to(device=dev, dtype=torch.bool).contiguous()
        L_act = 0
        if action_token is not None:
            L_act = action_token.shape[0]
            action_token = action_token.to(device=dev, dtype=torch.float32).contiguous()
        if prompt_token.stride(-1) != 1:
            prompt_token = prompt_token.contiguous()
        prompt_token = prompt_token.to(device=dev, dtype=torch.float32)
        prompt_token_mask = prompt_token_mask.to(device=dev, dtype=torch.bool).contiguous()
        Lp = prompt_token.shape[0]
        out = torch.empty(L_obs, B, E, dtype=torch.float32, device=dev)
        mode = 0
        key = self._prompt_key(prompt_token, prompt_token_mask) if self.cache_prompt_kv else None
        if key is not None:
            mode = 2 if key == self._kv_key else 1
            self._kv_key = None   # invalid until the call below succeeded
        _lib.check(self._lib.vima_decode(
            self._handle, _ptr(obs_token), _ptr(obs_mask), _ptr(action_token), L_obs, B, Q, L_act, _ptr(prompt_token),
            prompt_token.stride(1), prompt_token.stride(0), _ptr(prompt_token_mask), Lp, mode, _ptr(out), self._stream()))
        if key is not None:
            self._kv_key = key
            self._kv_keepalive = (prompt_token, prompt_token_mask)   # the key is only meaningful while these are alive
        return out

    @staticmethod
    def _prompt_key(prompt_token, prompt_token_mask):
        """Identity of the prompt for the cross-step K/V cache: storage pointer, in-place version counter, layout.
        Inference tensors (created under torch.inference_mode()) do not track a version counter: no key -> the call is
        stateless (mode 0), exactly like the reference."""
        if prompt_token.is_inference() or prompt_token_mask.is_inference():
            return None
        try:
            return (prompt_token.data_ptr(), prompt_token._version, tuple(prompt_token.shape), tuple(prompt_token.stride()),
                    prompt_token_mask.data_ptr(), prompt_token_mask._version)
        except RuntimeError:
            return None

    def reset_prompt_cache(self):
        """Forget the cached per-layer prompt K/V. CONTRACT of the default-on cache (`cache_prompt_kv`): `forward` reuses
        the K/V projections of the previous call when it is handed the same prompt tensor object state (same storage,
        same torch version counter, same layout). Writers that bypass the version counter (`.data.copy_`, DLPack / raw
        pointer writers, external kernels) must call this -- or start every episode with it -- otherwise stale K/V would
        be used. Set `policy.cache_prompt_kv = False` for the reference's stateless behaviour."""
        self._kv_key = None
        self._kv_keepalive = None

    def new_episode(self):
        """Episode-start hook: drops the prompt K/V cache (see reset_prompt_cache)."""
        self.reset_prompt_cache()

    def forward_step(self, obs_token: torch.Tensor, obs_mask: torch.Tensor, prev_action_token: torch.Tensor | None,
                     prompt_token: torch.Tensor, prompt_token_mask: torch.Tensor, step: int):
        """Incremental decoding for the env-step loop (no reference counterpart: scripts/example.py re-feeds the whole
        history to `forward` every step). Pass only the tokens of env step `step`: obs_token [B,Q,E] (or [1,B,Q,E]),
        obs_mask [B,Q] and, for step > 0, the embedded action of the previous step [B,E] (or [1,B,E]). step 0 starts
        an episode (prompt K/V and history caches live in the native handle). Returns the predicted action token
        [B,E] of this step == forward(<full history>)[step]."""
        self._ready()
        dev = self._device
        if obs_token.dim() == 4:
            obs_token, obs_mask = obs_token[-1], obs_mask[-1]
        B, Q, E = obs_token.shape
        obs_token = obs_token.to(device=dev, dtype=torch.float32).contiguous()
        obs_mask = obs_mask.to(device=dev, dtype=torch.bool).contiguous()
        if step == 0:
            assert torch.all(obs_mask[:, 0]), "first observation token of every sample must be valid (position id >= 0)"
        if prev_action_token is not None:
            if prev_action_token.dim() == 3:
                prev_action_token = prev_action_token[-1]
            prev_action_token = prev_action_token.to(device=dev, dtype=torch.float32).contiguous()
        if prompt_token.stride(-1) != 1:
            prompt_token = prompt_token.contiguous()
        prompt_token = prompt_token.to(dev)
        prompt_token_mask = prompt_token_mask.to(device=dev, dtype=torch.bool).contiguous()
        Lp = prompt_token.shape[0]
        out = torch.empty(B, E, dtype=torch.float32, device=dev)
        self._kv_key = None   # the native prompt cache now belongs to this episode
        _lib.check(self._lib.vima_decode_step(
            self._handle, _ptr(obs_token), _ptr(obs_mask), _ptr(prev_action_token), int(step), B, Q, _ptr(prompt_token),
            prompt_token.stride(1), prompt_token.stride(0), _ptr(prompt_token_mask), Lp, _ptr(out), self._stream()))
        return out

    def restart_samples(self, restart, prompt_token: torch.Tensor, prompt_token_mask: torch.Tensor):
        """Per-sample episode restart inside a batch that is stepping with `forward_step` (batched environments finish their
        episodes at different steps): for every sample with `restart[b]` true its cached history is forgotten, its position ids
        restart at 0, its next `forward_step` ignores the previous-action token, and its prompt K/V cache rows are rebuilt from
        `prompt_token[:, b]` / `prompt_token_mask[b]` (the new episode's prompt, same [Lp, B, E] / [B, Lp] layout as `forward_step`;
        other samples' rows are not read). Keep calling `forward_step(..., step=previous + 1)` with the UPDATED prompt tensors."""
        self._ready()
        dev = self._device
        flags = torch.as_tensor(restart).to(dtype=torch.uint8, device="cpu").contiguous()
        if prompt_token.stride(-1) != 1:
            prompt_token = prompt_token.contiguous()
        prompt_token = prompt_token.to(dev)
        prompt_token_mask = prompt_token_mask.to(device=dev, dtype=torch.bool).contiguous()
        Lp, B = prompt_token.shape[0], prompt_token.shape[1]
        assert flags.numel() == B
        self._kv_key = None
        _lib.check(self._lib.vima_decode_restart(self._handle, ctypes.c_void_p(flags.data_ptr()), B, _ptr(prompt_token),
                                                 prompt_token.stride(1), prompt_token.stride(0), _ptr(prompt_token_mask), Lp, self._stream()))

    # ------------------------------------------------------------------ actions
    def action_logits(self, predicted_action_tokens: torch.Tensor) -> torch.Tensor:
        """Raw concatenated logits [..., 700] of the 12 action-head MLPs (input of MultiCategoricalHead,
        action_decoder.py:165-166)."""
        self._ready()
        lead = predicted_action_tokens.shape[:-1]
        t = predicted_action_tokens.to(device=self._device, dtype=torch.float32).reshape(-1, self.embed_dim).contiguous()
        out = torch.empty(t.shape[0], N_LOGITS, dtype=torch.float32, device=self._device)
        _lib.check(self._lib.vima_action_head(self._handle, _ptr(t), t.shape[0], _ptr(out), self._stream()))
        return out.view(*lead, N_LOGITS)

    def forward_action_decoder(self, predicted_action_tokens: torch.Tensor):
        """-> {key: MultiCategorical} (vima_policy.py:264-265, action_decoder.py:51-52)."""
        raw = self.action_logits(predicted_action_tokens)
        out, off = {}, 0
        for k, dims in ACTION_DIMS.items():
            w = sum(dims)
            out[k] = MultiCategorical(raw[..., off:off + w], dims)
            off += w
        return out

    # the name BASELINE.json's north_star uses for the same step (the reference itself has no such symbol)
    discrete_action_head = forward_action_decoder

    def forward_action_token(self, action):
        """{key: int64 bin indices [..., 2|4]} -> action tokens [..., E] (vima_policy.py:261-262)."""
        self._ready()
        if set(action.keys()) != set(ACTION_KEYS):   # action_embd.py:30-32
            raise AssertionError(f"expected action keys {ACTION_KEYS}, got {sorted(action.keys())}")
        lead = action[ACTION_KEYS[0]].shape[:-1]
        idx = [action[k].to(device=self._device, dtype=torch.int64).reshape(-1, action[k].shape[-1]).contiguous()
               for k in ACTION_KEYS]
        R = idx[0].shape[0]
        out = torch.empty(R, self.embed_dim, dtype=torch.float32, device=self._device)
        arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in idx])
        _lib.check(self._lib.vima_action_embed(self._handle, arr, R, _ptr(out), self._stream()))
        return out.view(*lead, self.embed_dim)

    def discretize_action(self, action):
        """vima_policy.py:267-299 (training-side helper; mutates `action` like the reference)."""
        device = action["pose0_position"].device
        bx = torch.linspace(0, 1, self._n_discrete_x_bins, device=device)
        by = torch.linspace(0, 1, self._n_discrete_y_bins, device=device)
        br = torch.linspace(0, 1, self._n_discrete_rot_bins, device=device)
        for k in ("pose0_position", "pose1_position"):
            action[k][..., 0] = torch.bucketize(action[k][..., 0].contiguous(), bx)
            action[k][..., 1] = torch.bucketize(action[k][..., 1].contiguous(), by)
        for k in ("pose0_rotation", "pose1_rotation"):
            action[k] = torch.bucketize(action[k].contiguous(), br)
        return {k: v.long() for k, v in action.items()}

    def _de_discretize_actions(self, actions):
        """vima_policy.py:301-322."""
        actions = {k: v.float() for k, v in actions.items()}
        for k in ("pose0_position", "pose1_position"):
            actions[k][..., 0] = actions[k][..., 0] / self._n_discrete_x_bins
            actions[k][..., 1] = actions[k][..., 1] / self._n_discrete_y_bins
        for k in ("pose0_rotation", "pose1_rotation"):
            actions[k] = actions[k] / self._n_discrete_rot_bins
        return actions

    # ------------------------------------------------------------------ instrumentation
    def graph_stats(self):
        """(replays, captures) of the hipGraph replay mode (set_option("graphs", 1))."""
        self._ready()
        r, c = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self._lib.vima_graph_stats(self._handle, ctypes.byref(r), ctypes.byref(c)))
        return int(r.value), int(c.value)

    def prof_enable(self, on: bool = True):
        self._ensure_handle()
        _lib.check(self._lib.vima_prof_enable(self._handle, 1 if on else 0))

    def prof_read(self):
        """-> {"gemm"|"attention"|"other": {"ms","launches","flops"}} measured with HIP events on the launch stream."""
        ms = (ctypes.c_double * 3)()
        n = (ctypes.c_int64 * 3)()
        fl = (ctypes.c_double * 3)()
        _lib.check(self._lib.vima_prof_read(self._handle, ms, n, fl))
        names = ("gemm", "attention", "other")
        return {names[i]: {"ms": ms[i], "launches": int(n[i]), "flops": fl[i]} for i in range(3)}

    def prof_read_ex(self):
        """Like prof_read with the GEMMs split into "gemm" (no residual epilogue) and "gemm_residual" (fp32-residual
        epilogue) and the algorithmic HBM bytes of the GEMM launches."""
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_int64 * 4)()
        fl = (ctypes.c_double * 4)()
        by = (ctypes.c_double * 4)()
        _lib.check(self._lib.vima_prof_read_ex(self._handle, ms, n, fl, by))
        names = ("gemm", "attention", "other", "gemm_residual")
        return {names[i]: {"ms": ms[i], "launches": int(n[i]), "flops": fl[i], "bytes": by[i]} for i in range(4)}

    _GEMM_KINDS = {1: "vima::gemm_pp_kernel", 2: "vima::gemm_persistent_kernel", 3: "vima::gemm_wide_kernel",
                   4: "vima::gemm_kernel<Tile<256, 256>>", 5: "vima::gemm_kernel<Tile<128, 128>>", 6: "vima::gemm_kernel<Tile<64, 64>>",
                   7: "vima::gemm_kernel<Tile<32, 64>>", 8: "vima::gemm_kernel (two-pass split-K)", 9: "vima::gemm_pp_kernel",
                   10: "vima::gemm_resident_kernel<RTile<32, 32, 1, 1, 4>>", 11: "vima::gemm_resident_kernel<RTile<64, 32, 2, 1, 2>>",
                   12: "vima::gemm_resident_kernel<RTile<64, 64, 2, 2, 2>>", 15: "vima::gemm_resident_kernel<RTile<32, 32, 1, 1, 2, true>> (GEGLU pair)",
                   16: "vima::gemm_resident_kernel<RTile<64, 64, 2, 2, 1, true>> (GEGLU pair)",
                   17: "vima::gemm_skinny_kernel", 18: "vima::gemm_skinny_kernel (GEGLU pair)", 19: "vima::gemm_q4_kernel", 20: "vima::gemm_q4_kernel"}

    def _gemm_kernel_name(self, kid: int) -> str:
        kind, rest = divmod(int(kid), 1000)
        act, epi = divmod(rest, 10)
        base = self._GEMM_KINDS.get(kind, f"gemm kind {kind}")
        if kind == 9:
            return f"{base}<{act - 1}, {epi}, true>"          # fp8 e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4)
        if kind == 1:
            return f"{base}<{act - 1}, {epi}, false>"
        if kind in (19, 20):
            return f"{base}<{act - 1}, {epi}, {2 if kind == 19 else 1}>"       # <ACT, EPI, MIH>: 256x384 / 128x384 tile
        return f"{base}<{act - 1}, {epi}>" if kind in (2, 3) else f"{base} act {act - 1}"

    def prof_read_gemm_launches(self):
        """GEMM launches recorded since prof_enable(True), ONE BY ONE in launch order (call BEFORE prof_read / prof_read_ex) ->
        [{"kernel": rocprofv3 name, "M", "N", "K", "us"}]."""
        n = self._lib.vima_prof_read_gemm_launches(self._handle, 0, None, None, None)
        if n < 0:
            _lib.check(1)
        ids = (ctypes.c_int32 * max(n, 1))()
        mnk = (ctypes.c_int32 * (3 * max(n, 1)))()
        us = (ctypes.c_float * max(n, 1))()
        if self._lib.vima_prof_read_gemm_launches(self._handle, n, ids, mnk, us) < 0:
            _lib.check(1)
        return [{"kernel": self._gemm_kernel_name(ids[i]), "M": int(mnk[3 * i]), "N": int(mnk[3 * i + 1]), "K": int(mnk[3 * i + 2]),
                 "us": float(us[i])} for i in range(n)]

    def prof_read_gemm_kernels(self):
        """GEMM launches recorded since prof_enable(True), grouped by the kernel the launcher chose (call BEFORE prof_read /
        prof_read_ex, which reset the records) -> {kernel name as rocprofv3 prints it: {ms, launches, flops, bytes}}."""
        n = 64
        ids = (ctypes.c_int32 * n)()
        ms = (ctypes.c_double * n)()
        ln = (ctypes.c_int64 * n)()
        fl = (ctypes.c_double * n)()
        by = (ctypes.c_double * n)()
        k = self._lib.vima_prof_read_gemm_kernels(self._handle, n, ids, ms, ln, fl, by)
        if k < 0:
            _lib.check(1)
        out = {}
        for i in range(k):
            name = self._gemm_kernel_name(ids[i])
            out[name] = {"ms": ms[i], "launches": int(ln[i]), "flops": fl[i], "bytes": by[i]}
        return out

    def fp8_act_scales(self, group: str = "t5"):
        """precision "fp8": the calibrated activation scales of a group -- "t5" [12, 4], "vit" [4, 4] or "kv" [1] -- or None before
        that group's calibrating pass."""
        buf = (ctypes.c_float * 64)()
        n = self._lib.vima_fp8_act_scales(self._handle, {"t5": 0, "vit": 1, "kv": 2}[group], buf, 64)
        if n < 0:
            _lib.check(1)
        if n == 0:
            return None
        t = torch.tensor(list(buf[:n]))
        return t.view(-1, 4) if group != "kv" else t

    def workspace_bytes(self) -> int:
        return int(self._lib.vima_workspace_bytes(self._handle)) if self._handle is not None else 0
