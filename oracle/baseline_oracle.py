"""CPU ORACLE of the reference's baseline policies -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as
oracle/vima_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

Restates, in plain fp32 torch ops, VIMAGPTPolicy (vima/policy/vima_gpt_policy.py), VIMAGatoPolicy
(vima/policy/vima_gato_policy.py) and VIMAFlamingoPolicy (vima/policy/vima_flamingo_policy.py) with their encoders
(vima/nn/obj_encoder/obj_encoder.py:98-246, vit/vit.py:49-135,239-329, perceiver/perceiver.py) and the decoder-only
HFGPT (vima/nn/seq_modeling/gpt/gpt.py). Third-party arithmetic restated because it is not under /root/reference:
HF `modeling_perceiver.{PerceiverSelfAttention,PerceiverAttention,PerceiverMLP,PerceiverLayer,PerceiverEncoder}`
(transformers, unpinned in requirements.txt; installed 5.15.0) with the PerceiverConfig defaults the reference leaves
untouched (qk_channels = v_channels = None, cross_attention_shape_for_attention "kv", widening factors 1, gelu,
use_query_residual, layer_norm_eps 1e-12 unused by nn.LayerNorm's default 1e-5).

Parity pinning: tests/golden/baseline_*.npz hold OUTPUTS OF THE UNMODIFIED REFERENCE MODULES (oracle/make_golden.py,
imported through oracle/ref_shim.py, seeded weights loaded strict=True); tests/test_baseline_oracle.py checks this file
against them and re-checks live when /root/reference is present.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .vima_oracle import OraclePolicy, VIEWS, IMG_MEAN, IMG_STD, FMIN, _lin, _mlp


class _BaselineOracle(OraclePolicy):
    kind = None
    n_queries = 1            # `_obj_xf_num_queries`: tokens one image contributes

    def __init__(self, state_dict, *, embed_dim, n_layer, n_head, xattn_n_heads=None):
        super().__init__(state_dict, embed_dim=embed_dim, xf_n_layers=n_layer, sattn_n_heads=n_head,
                         xattn_n_heads=xattn_n_heads or n_head)

    # ------------------------------------------------------------------ image encoders
    def _vit_rect(self, img_u8, cls):
        """(Gato)ViTEncoder(Rectangular).forward (vit.py:49-81, 239-268) + basic_image_tensor_preprocess +
        VisionTransformerRectangular.forward (vit.py:305-329; cls=True -> ln_post(x[:, 0]) @ projection, [n, E]) or
        GatoVisionTransformerRectangular.forward (vit.py:119-135; cls=False -> ln_post(x) @ projection, [n, 8, E])."""
        sd, p = self.sd, "obj_encoder.cropped_img_encoder.vit."
        x = img_u8.float().reshape(-1, 3, img_u8.shape[-2], img_u8.shape[-1])
        assert x.max() > 2, "img should be between [0, 255] before normalize"
        mean = torch.tensor(IMG_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
        std = torch.tensor(IMG_STD, dtype=torch.float32).view(1, 3, 1, 1)
        x = (x / 255.0 - mean) / std
        x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=sd[p + "conv1.weight"].shape[-1])
        M = x.shape[0]
        x = x.reshape(M, x.shape[1], -1).permute(0, 2, 1)                      # [M, 8, 768], patches row-major
        if cls:
            x = torch.cat([sd[p + "cls_token"].view(1, 1, -1).expand(M, 1, -1), x], dim=1)
        x = x + sd[p + "pos_embed"]
        W = x.shape[-1]
        x = F.layer_norm(x, (W,), sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], 1e-5)
        heads = 24
        d = W // heads
        n_blocks = 1 + max(int(k[len(p + "blocks."):].split(".")[0]) for k in sd if k.startswith(p + "blocks."))
        for j in range(n_blocks):
            b = f"{p}blocks.{j}."
            h = F.layer_norm(x, (W,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5)
            q, k, v = _lin(h, sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"]).split(W, dim=-1)
            q = q.view(M, -1, heads, d).transpose(1, 2)
            k = k.view(M, -1, heads, d).transpose(1, 2)
            v = v.view(M, -1, heads, d).transpose(1, 2)
            att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
            x = x + _lin(att.transpose(1, 2).reshape(M, -1, W), sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"])
            h = F.layer_norm(x, (W,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5)
            h = _lin(h, sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"])
            h = h * torch.sigmoid(1.702 * h)
            x = x + _lin(h, sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"])
        if cls:
            x = x[:, 0, :]
        x = F.layer_norm(x, (W,), sd[p + "ln_post.weight"], sd[p + "ln_post.bias"], 1e-5)
        return x @ sd[p + "projection"]

    def obj_encoder(self, rgb):
        raise NotImplementedError

    # ------------------------------------------------------------------ prompt / obs
    def forward_prompt_assembly(self, prompts):
        """vima_gpt_policy.py:189-247 / vima_gato_policy.py:190-251 / vima_flamingo_policy.py:156-214: every image
        contributes `n_queries` tokens, there are no object masks (all assembled tokens are valid)."""
        sd = self.sd
        token_types, word_batch, image_batch = prompts
        word_emb = sd["prompt_embedding._embed_layer.weight"][word_batch]
        img_emb = self.obj_encoder(image_batch["rgb"])
        img_emb = _mlp(sd, "prompt_obj_post_layer", img_emb, 3)
        if img_emb.dim() == 2:
            img_emb = img_emb.unsqueeze(1)                                       # gpt: one token per image
        Q = self.n_queries
        lens = []
        for p in token_types:
            for t in p:
                if t not in (0, 1):
                    raise ValueError(f"Invalid prompt token type {t}")
            lens.append(sum(1 if t == 0 else Q for t in p))
        L_max, B = max(lens), len(token_types)
        toks = torch.zeros(B, L_max, 768, dtype=torch.float32)
        masks = torch.zeros(B, L_max, dtype=torch.bool)
        wp = ip = 0
        for b, p in enumerate(token_types):
            l = 0
            for t in p:
                if t == 0:
                    toks[b, l] = word_emb[wp]
                    wp += 1
                    l += 1
                else:
                    toks[b, l:l + Q] = img_emb[ip]
                    ip += 1
                    l += Q
            masks[b, :l] = True
        out = self._t5(toks, masks.float())
        if "t5_prompt_encoder_post_layer.weight" in sd:
            out = _lin(out, sd["t5_prompt_encoder_post_layer.weight"])
        return out.transpose(0, 1), masks

    def forward_obs_token(self, obs):
        """vima_gpt_policy.py:249-259 ([T,B,E]); vima_gato_policy.py:253-264 / vima_flamingo_policy.py:216-227
        ([T,B,Q,E], the end-effector feature repeated over the Q tokens)."""
        sd = self.sd
        rgb, ee = obs["rgb"], obs["ee"]
        lead = ee.shape[:2]
        feats = self.obj_encoder({v: rgb[v].reshape(-1, *rgb[v].shape[2:]) for v in VIEWS})
        feats = feats.reshape(*lead, *feats.shape[1:])
        ee_f = sd["end_effector_encoder.weight"][ee]
        if feats.dim() == 4:
            ee_f = ee_f.unsqueeze(2).expand(-1, -1, feats.shape[2], -1)
        return _lin(torch.cat([feats, ee_f], dim=-1), sd["obs_fusion_layer.weight"], sd["obs_fusion_layer.bias"])

    # ------------------------------------------------------------------ decoder-only GPT
    def _hfgpt(self, tokens, key_mask, position_ids):
        """HFGPT.forward (gpt/gpt.py:45-80) -> OpenAIGPTModel.forward (:113-220): inputs_embeds + positions_embed, then the
        post-LN blocks with the causal fill and the additive key mask. Batch-first [B, L, E]."""
        sd = self.sd
        x = tokens + sd["transformer.lm.positions_embed.weight"][position_ids]
        key_add = (1.0 - key_mask[:, None, None, :].float()) * FMIN
        for i in range(self.n_layers):
            x = self._block(i, x, key_add, prefix="transformer.lm.h.")
        return x

    def _decoder_only_forward(self, obs_token, action_token, prompt_token, prompt_token_mask):
        """vima_gpt_policy.py:118-187 (Q = 1, obs_token [T,B,E]) / vima_gato_policy.py:115-188 (Q = 16, [T,B,Q,E])."""
        sd, Q, E = self.sd, self.n_queries, self.embed_dim
        if obs_token.dim() == 3:
            obs_token = obs_token.unsqueeze(2)
        T, B = obs_token.shape[:2]
        Lp = prompt_token.shape[0]
        L_act = 0 if action_token is None else action_token.shape[0]
        L = T * Q + L_act + Lp + 1
        tokens = torch.zeros(L, B, E, dtype=torch.float32)
        tokens[:Lp] = prompt_token
        tokens[Lp] = sd["prompt_sep_token"]
        for t in range(T):
            s = Lp + 1 + t * (Q + 1)
            tokens[s:s + Q] = obs_token[t].transpose(0, 1)
            if t < L_act:
                tokens[s + Q] = action_token[t]
        mask = torch.cat([prompt_token_mask, torch.ones(B, L - Lp, dtype=torch.bool)], dim=1)
        nv = prompt_token_mask.sum(dim=1)
        pos = torch.zeros(B, L, dtype=torch.long)
        for b in range(B):
            n = int(nv[b])
            pos[b, :n] = torch.arange(n)
            pos[b, n:Lp] = n - 1
            pos[b, Lp:] = torch.arange(n, n + L - Lp)
        out = self._hfgpt(tokens.transpose(0, 1), mask, pos).transpose(0, 1)
        return out[Lp + 1 + Q - 1::Q + 1]


class OracleGPTPolicy(_BaselineOracle):
    kind, n_queries = "gpt", 1

    def obj_encoder(self, rgb):
        """MultiViewRGBEncoder.forward (obj_encoder.py:232-240): cls features of the views concatenated on the feature axis."""
        return torch.cat([self._vit_rect(rgb[v], True) for v in VIEWS], dim=-1)

    def forward(self, obs_token, action_token, prompt_token, prompt_token_mask):
        return self._decoder_only_forward(obs_token, action_token, prompt_token, prompt_token_mask)

    __call__ = forward


class OracleGatoPolicy(_BaselineOracle):
    kind, n_queries = "gato", 16

    def obj_encoder(self, rgb):
        """GatoMultiViewRGBEncoder.forward (obj_encoder.py:123-139): patch tokens of the views concatenated on the token axis."""
        return torch.cat([self._vit_rect(rgb[v], False) for v in VIEWS], dim=-2)

    def forward(self, obs_token, action_token, prompt_token, prompt_token_mask):
        return self._decoder_only_forward(obs_token, action_token, prompt_token, prompt_token_mask)

    __call__ = forward


class OracleFlamingoPolicy(_BaselineOracle):
    kind, n_queries = "flamingo", 4

    def _perceiver_layer(self, pre, heads, hidden, inputs=None):
        """PerceiverLayer.forward (modeling_perceiver.py:383-415) = PerceiverAttention (:306-332: LN(q) [, LN(kv)], q/k/v
        Linear, softmax(q k^T / sqrt(d)) v, output dense, + query residual) then hidden + MLP(LN(hidden)) (:347-351,
        :404-415; exact erf GELU). All inputs are valid here (obj_encoder.py:199-202: mask of ones -> additive 0)."""
        sd, E = self.sd, self.embed_dim
        a = pre + "attention."
        d = E // heads
        h = F.layer_norm(hidden, (E,), sd[a + "self.layernorm1.weight"], sd[a + "self.layernorm1.bias"], 1e-5)
        if inputs is None:
            kv = h
        else:
            kv = F.layer_norm(inputs, (E,), sd[a + "self.layernorm2.weight"], sd[a + "self.layernorm2.bias"], 1e-5)
        n = hidden.shape[0]
        q = _lin(h, sd[a + "self.query.weight"], sd[a + "self.query.bias"]).view(n, -1, heads, d).transpose(1, 2)
        k = _lin(kv, sd[a + "self.key.weight"], sd[a + "self.key.bias"]).view(n, -1, heads, d).transpose(1, 2)
        v = _lin(kv, sd[a + "self.value.weight"], sd[a + "self.value.bias"]).view(n, -1, heads, d).transpose(1, 2)
        ctx = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
        ctx = ctx.transpose(1, 2).reshape(n, -1, E)
        att = _lin(ctx, sd[a + "output.dense.weight"], sd[a + "output.dense.bias"]) + hidden
        m = F.layer_norm(att, (E,), sd[pre + "layernorm.weight"], sd[pre + "layernorm.bias"], 1e-5)
        m = F.gelu(_lin(m, sd[pre + "mlp.dense1.weight"], sd[pre + "mlp.dense1.bias"]))
        return _lin(m, sd[pre + "mlp.dense2.weight"], sd[pre + "mlp.dense2.bias"]) + att

    def obj_encoder(self, rgb):
        """MultiViewRGBPerceiverEncoder.forward (obj_encoder.py:195-204) -> PerceiverModel / PerceiverEncoder.forward
        (modeling_perceiver.py:468-527): one cross-attention of the 4 latents onto the 16 patch tokens, then 4 blocks x the
        SAME 4 self-attention layers (weights shared across blocks)."""
        sd, pc = self.sd, "obj_encoder.peceiver.model."
        feats = torch.cat([self._vit_rect(rgb[v], False) for v in VIEWS], dim=-2)            # [n, 16, E]
        z = sd[pc + "embeddings.latents"].unsqueeze(0).expand(feats.shape[0], -1, -1)
        z = self._perceiver_layer(pc + "encoder.cross_attention.", 8, z, feats)
        n_self = 1 + max(int(k[len(pc + "encoder.self_attends."):].split(".")[0]) for k in sd if k.startswith(pc + "encoder.self_attends."))
        for _ in range(4):
            for i in range(n_self):
                z = self._perceiver_layer(f"{pc}encoder.self_attends.{i}.", 8, z)
        return z

    def forward(self, obs_token, action_token, prompt_token, prompt_token_mask):
        """vima_flamingo_policy.py:121-154: the VIMAPolicy interleave with every token valid and default position ids."""
        mask = torch.ones(obs_token.shape[:3], dtype=torch.bool)
        return OraclePolicy.forward(self, obs_token, mask, action_token, prompt_token, prompt_token_mask)

    __call__ = forward


ORACLES = {"gpt": OracleGPTPolicy, "gato": OracleGatoPolicy, "flamingo": OracleFlamingoPolicy}


def build_baseline_oracle(cfg, state_dict):
    """cfg: vima_testing.synthetic.BaselineConfig"""
    return ORACLES[cfg.kind](state_dict, embed_dim=cfg.embed_dim, n_layer=cfg.n_layer, n_head=cfg.n_head,
                             xattn_n_heads=cfg.xattn_n_heads or None)
