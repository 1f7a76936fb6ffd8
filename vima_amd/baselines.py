"""Host-side mirrors of the reference's BASELINE policies, backed by the same gfx950 HIP library:

    VIMAGPTPolicy        vima/policy/vima_gpt_policy.py       (decoder-only HFGPT, one token per frame pair)
    VIMAGatoPolicy       vima/policy/vima_gato_policy.py      (decoder-only HFGPT, 16 patch tokens per frame pair)
    VIMAFlamingoPolicy   vima/policy/vima_flamingo_policy.py  (Perceiver-resampled tokens + XAttnGPT)

Same constructor arguments, method surface, return shapes and state_dict keys as the reference classes; the arithmetic runs
in the HIP kernels (`vima_rgb_obs_encode`, `vima_rgb_prompt_encode`, `vima_seq_decode` / `vima_decode`, `vima_action_*`).
No CPU fallback. Unlike VIMAPolicy these consume whole RGB frames `{"rgb": {view: u8 [..., 3, 64, 128]}}` and have no
object masks (`forward` takes no `obs_mask`)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .policy import VIMAPolicy, VIEWS, build_prompt_index, _ptr, _pair

RGB_SHAPE = (3, 64, 128)     # img_size=(64, 128) (vima_gpt_policy.py:37-45)


class _BaselinePolicy(VIMAPolicy):
    KIND = None

    def __init__(self, *, embed_dim, n_layer, n_head, xattn_n_heads=None, vocab_size=40478, n_positions=512,
                 precision="bf16", device=None):
        if precision in ("fp8w", "fp8"):
            raise ValueError(f"precision '{precision}' is validated for VIMAPolicy only; the baseline policies run in 'bf16' or 'fp32'")
        super().__init__(embed_dim=embed_dim, xf_n_layers=n_layer, sattn_n_heads=n_head,
                         xattn_n_heads=xattn_n_heads or n_head, xattn_n_positions=256, n_positions=n_positions,
                         precision=precision, device=device)
        self._cfg.policy_kind = _lib.POLICY_KIND[self.KIND]
        self._vocab_size = vocab_size
        self._obj_xf_num_queries = int(self._lib.vima_rgb_tokens_per_image(ctypes.byref(self._cfg)))
        self.cache_prompt_kv = self.KIND == "flamingo"

    # ------------------------------------------------------------------ weights
    def expected_keys(self):
        req = _lib.required_params(self._cfg)
        n = self._cfg.xf_n_layers
        ign = ["t5_prompt_encoder.t5.shared.weight", "t5_prompt_encoder.t5.encoder.embed_tokens.weight"]
        if self.KIND == "flamingo":
            ign += ["xattn_gpt.position_ids", "xattn_gpt.xattn_position_ids"]
            ign += [f"xattn_gpt.h.{i}.attn.bias" for i in range(n)]
            ign += [f"xattn_gpt.xattns.{i}.kv_position_ids" for i in range(n)]
        else:   # tokens_embed is never read: the policies feed inputs_embeds (gpt/gpt.py:69-73)
            ign += ["transformer.lm.position_ids", "transformer.lm.tokens_embed.weight"]
        return req, ign

    # ------------------------------------------------------------------ frames
    def _rgb_inputs(self, rgb, lead_dims):
        frames = [f.reshape(-1, *f.shape[lead_dims:]) for f in self._views_of(rgb, "rgb")]
        n = frames[0].shape[0]
        for f in frames:
            if tuple(f.shape) != (n, *RGB_SHAPE):
                raise ValueError(f"rgb must be [..., 3, 64, 128] for every view, got {tuple(f.shape)}")
        self._check_img(frames)
        return [f.to(device=self._device, dtype=torch.uint8).contiguous() for f in frames], n

    def obj_encoder(self, rgb):
        """obj_encoder.forward for ONE leading dim: gpt [n, 2E]; gato [n, 16, E]; flamingo [n, 4, E] (obj_encoder.py:123-240)."""
        self._ready()
        frames, n = self._rgb_inputs(rgb, 1)
        Q, E = self._obj_xf_num_queries, self.embed_dim
        Eo = 2 * E if self.KIND == "gpt" else E
        out = torch.empty(n, Q, Eo, dtype=torch.float32, device=self._device)
        _lib.check(self._lib.vima_rgb_encode(self._handle, _pair(frames[0].data_ptr(), frames[1].data_ptr()), n, _ptr(out),
                                             self._stream()))
        return out.view(n, Eo) if self.KIND == "gpt" else out

    def forward_obs_token(self, obs):
        """obs = {"rgb": {view: u8 [L_obs, B, 3, 64, 128]}, "ee": [L_obs, B] int64} -> [L_obs, B, E] (gpt) or
        [L_obs, B, Q, E] (gato / flamingo)   (vima_gpt_policy.py:249-259, vima_gato_policy.py:253-264)"""
        self._ready()
        rgb, ee = obs["rgb"], obs["ee"]
        lead = tuple(ee.shape[:2])
        frames, n = self._rgb_inputs(rgb, 2)
        assert n == lead[0] * lead[1]
        ee = ee.to(device=self._device, dtype=torch.int64).contiguous()
        Q, E = self._obj_xf_num_queries, self.embed_dim
        out = torch.empty(n, Q, E, dtype=torch.float32, device=self._device)
        _lib.check(self._lib.vima_rgb_obs_encode(self._handle, _pair(frames[0].data_ptr(), frames[1].data_ptr()), _ptr(ee), n,
                                                 _ptr(out), self._stream()))
        return out.view(*lead, E) if self.KIND == "gpt" else out.view(*lead, Q, E)

    def forward_prompt_assembly(self, prompts):
        """(raw_prompts_token_type, word_batch, {"rgb": {view: u8 [n_img, 3, 64, 128]}}) -> (prompt_tokens [L, B, E],
        prompt_masks [B, L]); every image contributes `_obj_xf_num_queries` tokens (vima_gato_policy.py:190-251)."""
        self._ready()
        raw_types, word_batch, image_batch = prompts
        n_img_total = sum(1 for p in raw_types for t in p if t == 1)
        dev = self._device
        frames, n_img = (self._rgb_inputs(image_batch["rgb"], 1) if n_img_total > 0 else (None, 0))
        Q = self._obj_xf_num_queries
        B = len(raw_types)
        src, L_max, wp, ip = build_prompt_index(raw_types, Q)
        if wp > word_batch.numel() or ip > n_img:
            raise IndexError("prompt token types reference more words / images than provided")
        word_batch = word_batch.to(device=dev, dtype=torch.int64).contiguous()
        tok_src = torch.from_numpy(src[:, :L_max].copy()).to(dev)
        out = torch.empty(B, L_max, self.embed_dim, dtype=torch.float32, device=dev)
        omask = torch.empty(B, L_max, dtype=torch.bool, device=dev)
        _lib.check(self._lib.vima_rgb_prompt_encode(
            self._handle, _ptr(word_batch), int(word_batch.numel()),
            _pair(frames[0].data_ptr(), frames[1].data_ptr()) if frames else _pair(0, 0), n_img, _ptr(tok_src), B, L_max,
            _ptr(out), _ptr(omask), self._stream()))
        return out.transpose(0, 1), omask

    # ------------------------------------------------------------------ decoder
    def forward(self, obs_token, action_token, prompt_token, prompt_token_mask):
        """-> predicted_action_tokens [L_obs, B, E] (vima_gpt_policy.py:118-187, vima_gato_policy.py:115-188,
        vima_flamingo_policy.py:121-154). Note the baseline signature: no obs_mask."""
        self._ready()
        if obs_token.dim() == 3:
            obs_token = obs_token.unsqueeze(2)
        L_obs, B, Q, E = obs_token.shape
        if Q != self._obj_xf_num_queries:
            raise AssertionError(f"obs_token must carry {self._obj_xf_num_queries} tokens per step, got {Q}")
        if obs_token.dtype != torch.float32 or prompt_token.dtype != torch.float32:
            raise AssertionError(f"obs_token / prompt_token must be float32, got {obs_token.dtype} / {prompt_token.dtype}")
        if prompt_token.dim() != 3 or prompt_token.shape[1] != B or prompt_token.shape[2] != E:
            raise AssertionError(f"prompt_token must be [Lp, {B}, {E}], got {tuple(prompt_token.shape)}")
        if tuple(prompt_token_mask.shape) != (B, prompt_token.shape[0]):
            raise AssertionError("prompt_token_mask shape does not match the prompt tokens")
        if self.KIND == "flamingo":
            ones = torch.ones(L_obs, B, Q, dtype=torch.bool, device=self._device)
            return VIMAPolicy.forward(self, obs_token, ones, action_token, prompt_token, prompt_token_mask)
        dev = self._device
        obs_token = obs_token.to(device=dev, dtype=torch.float32).contiguous()
        L_act = 0
        if action_token is not None:
            L_act = action_token.shape[0]
            action_token = action_token.to(device=dev, dtype=torch.float32).contiguous()
        if prompt_token.stride(-1) != 1:
            prompt_token = prompt_token.contiguous()
        prompt_token = prompt_token.to(device=dev, dtype=torch.float32)
        prompt_token_mask = prompt_token_mask.to(device=dev, dtype=torch.bool).contiguous()
        Lp = prompt_token.shape[0]
        # the reference builds the position ids on the host from the per-sample number of valid prompt tokens
        # (vima_gato_policy.py:156-184, one `.item()` per sample); a sample without any valid prompt token gets the id -1 there and the
        # embedding lookup raises. The C entry point is asynchronous and clamps instead, so the same refusal happens here.
        if not bool(prompt_token_mask.any(dim=1).all()):
            raise IndexError("index out of range in self (a sample has no valid prompt token: position id -1, vima_gato_policy.py:164)")
        out = torch.empty(L_obs, B, E, dtype=torch.float32, device=dev)
        _lib.check(self._lib.vima_seq_decode(
            self._handle, _ptr(obs_token), _ptr(action_token), L_obs, B, L_act, _ptr(prompt_token), prompt_token.stride(1),
            prompt_token.stride(0), _ptr(prompt_token_mask), Lp, _ptr(out), self._stream()))
        return out

    def forward_step(self, obs_token, prev_action_token, prompt_token, prompt_token_mask, step: int):
        """Incremental decoding of one env step (see VIMAPolicy.forward_step). VIMAFlamingoPolicy decodes with XAttnGPT, so the
        episode caches of `vima_decode_step` apply unchanged (every token valid); the decoder-only policies re-feed the history."""
        if self.KIND != "flamingo":
            raise NotImplementedError("incremental decoding needs the XAttnGPT episode caches: VIMAPolicy / VIMAFlamingoPolicy only")
        if obs_token.dim() == 4:
            obs_token = obs_token[-1]
        ones = torch.ones(obs_token.shape[:2], dtype=torch.bool, device=self._device)
        return VIMAPolicy.forward_step(self, obs_token, ones, prev_action_token, prompt_token, prompt_token_mask, step)


class VIMAGPTPolicy(_BaselinePolicy):
    """vima/policy/vima_gpt_policy.py:9-116"""
    KIND = "gpt"

    def __init__(self, *, embed_dim: int, vocab_size=40478, n_positions=512, n_layer=12, n_head=12, dropout: float = 0.1,
                 precision="bf16", device=None):
        super().__init__(embed_dim=embed_dim, n_layer=n_layer, n_head=n_head, vocab_size=vocab_size, n_positions=n_positions,
                         precision=precision, device=device)


class VIMAGatoPolicy(_BaselinePolicy):
    """vima/policy/vima_gato_policy.py:10-113"""
    KIND = "gato"

    def __init__(self, *, embed_dim: int, vocab_size=40478, n_positions=512, n_layer=12, n_head=12, dropout: float = 0.1,
                 precision="bf16", device=None):
        super().__init__(embed_dim=embed_dim, n_layer=n_layer, n_head=n_head, vocab_size=vocab_size, n_positions=n_positions,
                         precision=precision, device=device)


class VIMAFlamingoPolicy(_BaselinePolicy):
    """vima/policy/vima_flamingo_policy.py:9-119"""
    KIND = "flamingo"

    def __init__(self, *, embed_dim: int, dt_n_layers: int, dt_n_heads: int, xattn_n_heads: int, precision="bf16", device=None):
        super().__init__(embed_dim=embed_dim, n_layer=dt_n_layers, n_head=dt_n_heads, xattn_n_heads=xattn_n_heads,
                         precision=precision, device=device)


BASELINES = {"gpt": VIMAGPTPolicy, "gato": VIMAGatoPolicy, "flamingo": VIMAFlamingoPolicy}


def build_baseline(cfg, precision="bf16", device=None):
    """cfg: vima_testing.synthetic.BaselineConfig"""
    kw = cfg.ctor_kwargs()
    return BASELINES[cfg.kind](**kw, precision=precision, device=device)
