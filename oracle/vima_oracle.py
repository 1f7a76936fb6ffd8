"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A dependency-free (plain `torch` functional ops, fp32, no `transformers`, no `einops`,
no reference imports) restatement of the arithmetic of the reference VIMA policy hot path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product (`vima_amd/`) never does and fails loudly when its HIP library is missing.

Parity pinning: the reference ships no tests / golden vectors, so this
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF: `oracle/make_golden.py` imports the
unmodified reference modules from /root/reference (through `oracle/ref_shim.py`), loads the seeded
synthetic state dict with strict=True and stores its outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this file against those fixtures, and
`tests/test_oracle_vs_reference.py` re-checks live whenever /root/reference is present.

Every function cites the reference lines it restates (paths relative to /root/reference).
Third-party arithmetic restated here because it is not under /root/reference (dependency is
unpinned in requirements.txt:6; installed 5.15.0): HF `modeling_openai.Attention/Conv1D`,
HF `modeling_t5.{T5LayerNorm,T5DenseActDense,T5Attention.compute_bias,_relative_position_bucket}`,
`torch.nn.MultiheadAttention`, `torch.distributions.Categorical` logit normalisation.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

VIEWS = ("front", "top")
ACTION_KEYS = ("pose0_position", "pose0_rotation", "pose1_position", "pose1_rotation")
ACTION_DIMS = {"pose0_position": [50, 100], "pose0_rotation": [50] * 4,
               "pose1_position": [50, 100], "pose1_rotation": [50] * 4}
IMG_MEAN = (0.3471, 0.3429, 0.3383)   # vit.py:9
IMG_STD = (0.3011, 0.2961, 0.2956)    # vit.py:10
FMIN = torch.finfo(torch.float32).min


def _lin(x, w, b=None):
    """nn.Linear: y = x @ W[out,in]^T + b."""
    return F.linear(x, w, b)


def _conv1d(x, w, b):
    """HF Conv1D: weight is [in, out]; y = x @ W + b (pytorch_utils.py Conv1D.forward)."""
    return torch.addmm(b, x.reshape(-1, x.shape[-1]), w).view(*x.shape[:-1], w.shape[1])


def _mlp(sd, prefix, x, n_layers):
    """build_mlp: [Linear, Identity, ReLU] x depth + Linear at Sequential indices 0,3,6 (nn/utils.py:84-91)."""
    for i in range(n_layers):
        x = _lin(x, sd[f"{prefix}.{3 * i}.weight"], sd[f"{prefix}.{3 * i}.bias"])
        if i < n_layers - 1:
            x = torch.relu(x)
    return x


def t5_relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """HF modeling_t5._relative_position_bucket, bidirectional branch. rel = key_pos - query_pos."""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, n, large)


class OraclePolicy:
    """Same method surface as the reference `VIMAPolicy` (vima/policy/vima_policy.py:11-322),
    evaluated from a reference-layout state dict. Pure inference (eval mode: dropout = identity)."""

    def __init__(self, state_dict, *, embed_dim, xf_n_layers, sattn_n_heads, xattn_n_heads, **_unused):
        self.sd = {k: v for k, v in state_dict.items()}
        self.embed_dim = embed_dim
        self.n_layers = xf_n_layers
        self.sattn_heads = sattn_n_heads
        self.xattn_heads = xattn_n_heads
        if embed_dim % sattn_n_heads or embed_dim % xattn_n_heads:
            raise ValueError("embed_dim must be divisible by the head counts")  # components.py:120-123

    # ------------------------------------------------------------------ object encoder
    def _fq(self, group, idx, site, t):
        """Hook for activation fake-quantisation (oracle/fp8_quant.py overrides it): group "vit" (block idx; sites 0 ln_1 output,
        1 attention output, 2 ln_2 output, 3 QuickGELU hidden) or "kv" (the prompt entering the cross-attention key_value
        projection). Identity here."""
        return t

    def _vit(self, img_u8):
        """ViTEncoder.forward (vit.py:36-46) + basic_image_tensor_preprocess (preprocess.py:9-43)
        + VisionTransformer.forward (vit.py:171-191) + ResidualAttentionBlock (vit.py:199-236)."""
        sd, p = self.sd, "obj_encoder.cropped_img_encoder.vit."
        lead = img_u8.shape[:-3]
        x = img_u8.float().reshape(-1, 3, img_u8.shape[-2], img_u8.shape[-1])
        assert x.max() > 2, "img should be between [0, 255] before normalize"  # preprocess.py:28
        mean = torch.tensor(IMG_MEAN, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(IMG_STD, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
        x = (x / 255.0 - mean) / std                                           # preprocess.py:41, :83
        x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=16)               # vit.py:172
        M = x.shape[0]
        x = x.reshape(M, x.shape[1], -1).permute(0, 2, 1)                      # [M, 4, 768]
        cls = sd[p + "cls_token"].view(1, 1, -1).expand(M, 1, -1)
        x = torch.cat([cls, x], dim=1) + sd[p + "pos_embed"]                   # vit.py:176-179
        W = x.shape[-1]
        x = F.layer_norm(x, (W,), sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], 1e-5)
        heads = 24
        d = W // heads
        n_blocks = 1 + max(int(k[len(p + "blocks."):].split(".")[0]) for k in sd if k.startswith(p + "blocks."))
        for j in range(n_blocks):
            b = f"{p}blocks.{j}."
            h = self._fq("vit", j, 0, F.layer_norm(x, (W,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5))
            qkv = _lin(h, sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"])  # nn.MultiheadAttention
            q, k, v = qkv.split(W, dim=-1)
            q = q.view(M, -1, heads, d).transpose(1, 2)
            k = k.view(M, -1, heads, d).transpose(1, 2)
            v = v.view(M, -1, heads, d).transpose(1, 2)
            att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
            att = self._fq("vit", j, 1, att.transpose(1, 2).reshape(M, -1, W))
            x = x + _lin(att, sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"])
            h = self._fq("vit", j, 2, F.layer_norm(x, (W,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5))
            h = _lin(h, sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"])
            h = self._fq("vit", j, 3, h * torch.sigmoid(1.702 * h))            # QuickGELU vit.py:194-196
            x = x + _lin(h, sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"])
        x = F.layer_norm(x[:, 0, :], (W,), sd[p + "ln_post.weight"], sd[p + "ln_post.bias"], 1e-5)
        x = x @ sd[p + "projection"]                                           # vit.py:188-189
        return x.view(*lead, -1)

    def obj_encoder(self, cropped_img, bbox, mask=None):
        """ObjEncoder.forward (obj_encoder.py:66-95): out [..., n_objs * n_views, E]."""
        sd = self.sd
        feats = []
        for view in VIEWS:
            img = self._vit(cropped_img[view])
            norm = torch.tensor([256.0, 128.0, 128.0, 256.0], device=img.device)   # obj_encoder.py:12-13,80-85
            bb = bbox[view].float() / norm
            bb = _mlp(sd, f"obj_encoder.bbox_mlp.{view}", bb, 3)
            feats.append(_lin(torch.cat([img, bb], dim=-1),
                              sd[f"obj_encoder.pre_transformer_layer.{view}.weight"],
                              sd[f"obj_encoder.pre_transformer_layer.{view}.bias"]))
        return torch.cat(feats, dim=-2)

    # ------------------------------------------------------------------ T5 prompt encoder
    def _t5(self, x, mask_f):
        """T5PromptEncoder.forward (prompt_encoder.py:30-58) -> T5Stack.forward (:212-473) ->
        T5Block (:491-604) -> T5Attention.forward (:682-825); x [B, L, 768], mask_f [B, L] float."""
        sd, p = self.sd, "t5_prompt_encoder.t5.encoder."
        B, L, D = x.shape
        H, dk = 12, 64
        ext = (1.0 - mask_f[:, None, None, :]) * FMIN            # get_extended_attention_mask
        pos = torch.arange(L, device=x.device)
        rel = pos[None, :] - pos[:, None]                        # memory - context (compute_bias)
        bucket = t5_relative_position_bucket(rel)
        rb = sd[p + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"][bucket]  # [L, L, H]
        position_bias = rb.permute(2, 0, 1).unsqueeze(0) + ext   # prompt_encoder.py:783-797
        n_layers = 1 + max(int(k[len(p + "block."):].split(".")[0]) for k in sd if k.startswith(p + "block."))
        for l in range(n_layers):
            a = f"{p}block.{l}.layer.0."
            h = self._rms(x, sd[a + "layer_norm.weight"])
            q = _lin(h, sd[a + "SelfAttention.q.weight"]).view(B, L, H, dk).transpose(1, 2)
            k = _lin(h, sd[a + "SelfAttention.k.weight"]).view(B, L, H, dk).transpose(1, 2)
            v = _lin(h, sd[a + "SelfAttention.v.weight"]).view(B, L, H, dk).transpose(1, 2)
            scores = q @ k.transpose(3, 2)                       # no 1/sqrt(d): prompt_encoder.py:771-773
            scores = scores + position_bias
            w = torch.softmax(scores.float(), dim=-1)
            o = (w @ v).transpose(1, 2).reshape(B, L, H * dk)
            x = x + _lin(o, sd[a + "SelfAttention.o.weight"])
            f = f"{p}block.{l}.layer.1."
            h = self._rms(x, sd[f + "layer_norm.weight"])
            h = torch.relu(_lin(h, sd[f + "DenseReluDense.wi.weight"]))
            x = x + _lin(h, sd[f + "DenseReluDense.wo.weight"])
        return self._rms(x, sd[p + "final_layer_norm.weight"])

    @staticmethod
    def _rms(x, w, eps=1e-6):
        """HF T5LayerNorm: no mean subtraction, no bias, fp32 variance."""
        var = x.float().pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + eps))

    def forward_prompt_assembly(self, prompts):
        """VIMAPolicy.forward_prompt_assembly (vima_policy.py:161-240)."""
        sd = self.sd
        token_types, word_batch, image_batch = prompts
        word_emb = sd["prompt_embedding._embed_layer.weight"][word_batch]          # word_embd.py:18-23
        img_emb = self.obj_encoder(image_batch["cropped_img"], image_batch["bbox"])
        img_emb = _mlp(sd, "prompt_obj_post_layer", img_emb, 3)                    # :165
        Q = img_emb.shape[-2]
        L_max = 0
        for p in token_types:
            n = 0
            for t in p:
                if t == 0:
                    n += 1
                elif t == 1:
                    n += Q
                else:
                    raise ValueError(f"Invalid prompt token type {t}")
            L_max = max(L_max, n)
        dev = word_emb.device
        B = len(token_types)
        toks = torch.zeros(B, L_max, 768, dtype=torch.float32, device=dev)
        masks = torch.zeros(B, L_max, dtype=torch.bool, device=dev)
        wp = ip = 0
        for b, p in enumerate(token_types):
            l = 0
            for t in p:
                if t == 0:
                    toks[b, l] = word_emb[wp]
                    masks[b, l] = True
                    wp += 1
                    l += 1
                else:
                    toks[b, l:l + Q] = img_emb[ip]
                    masks[b, l:l + Q] = torch.cat([image_batch["mask"][v][ip] for v in VIEWS], dim=-1)
                    ip += 1
                    l += Q
        out = self._t5(toks, masks.float())
        if "t5_prompt_encoder_post_layer.weight" in sd:                            # :97-101
            out = _lin(out, sd["t5_prompt_encoder_post_layer.weight"])
        return out.transpose(0, 1), masks

    # ------------------------------------------------------------------ observations / actions
    def forward_obs_token(self, obs):
        """VIMAPolicy.forward_obs_token (vima_policy.py:242-259)."""
        sd = self.sd
        objects, ee = obs["objects"], obs["ee"]
        lead = ee.shape[:2]
        crops = {v: objects["cropped_img"][v].reshape(-1, *objects["cropped_img"][v].shape[2:]) for v in VIEWS}
        bbox = {v: objects["bbox"][v].reshape(-1, *objects["bbox"][v].shape[2:]) for v in VIEWS}
        feats = self.obj_encoder(crops, bbox)
        feats = feats.reshape(*lead, *feats.shape[1:])
        ee_f = sd["end_effector_encoder.weight"][ee]                               # Embedding(2,2)
        ee_f = ee_f.unsqueeze(2).expand(-1, -1, feats.shape[-2], -1)
        out = _lin(torch.cat([feats, ee_f], dim=-1), sd["obs_fusion_layer.weight"], sd["obs_fusion_layer.bias"])
        mask = torch.cat([objects["mask"][v].reshape(*lead, -1) for v in VIEWS], dim=-1)
        return out, mask

    def _de_discretize_actions(self, actions):
        """vima_policy.py:301-322 (x bins 50, y bins 100, rot bins 50)."""
        out = {k: v.float().clone() for k, v in actions.items()}
        for k in ("pose0_position", "pose1_position"):
            out[k][..., 0] = out[k][..., 0] / 50
            out[k][..., 1] = out[k][..., 1] / 100
        for k in ("pose0_rotation", "pose1_rotation"):
            out[k] = out[k] / 50
        return out

    def forward_action_token(self, action):
        """vima_policy.py:261-262 -> ActionEmbedding.forward (action_embd.py:29-37)."""
        sd = self.sd
        a = self._de_discretize_actions(action)
        parts = [_mlp(sd, f"action_encoder._embed_dict.{k}._layer", a[k], 2) for k in sorted(a.keys())]
        x = torch.cat(parts, dim=-1)
        if "action_encoder._post_layer.weight" not in sd:   # nn.Identity when embed_dim == 4 * 256 (action_embd.py:16-20)
            return x
        return _lin(x, sd["action_encoder._post_layer.weight"], sd["action_encoder._post_layer.bias"])

    def action_logits(self, tokens):
        """Raw concatenated MLP outputs, width 700, key order of ACTION_KEYS (action_decoder.py:165-166)."""
        sd = self.sd
        outs = []
        for k in ACTION_KEYS:
            for j in range(len(ACTION_DIMS[k])):
                outs.append(_mlp(sd, f"action_decoder._decoders.{k}.mlps.{j}", tokens, 3))
        return torch.cat(outs, dim=-1)

    def forward_action_decoder(self, tokens):
        """vima_policy.py:264-265; returns {key: {"logits": normalised per-dim log-probs list,
        "mode": int64 [..., n]}} (dists.py:12-28; Categorical(logits=) subtracts logsumexp)."""
        raw = self.action_logits(tokens)
        out, off = {}, 0
        for k in ACTION_KEYS:
            dims = ACTION_DIMS[k]
            w = sum(dims)
            chunk = raw[..., off:off + w]
            off += w
            splits = torch.split(chunk, dims, dim=-1)
            norm = [s - s.logsumexp(dim=-1, keepdim=True) for s in splits]
            mode = torch.stack([torch.softmax(s, dim=-1).argmax(dim=-1) for s in splits], dim=-1)
            out[k] = {"raw": chunk, "logits": norm, "mode": mode}
        return out

    # ------------------------------------------------------------------ decoder
    def _xattention(self, i, q, kv, mask):
        """XAttention.forward (components.py:158-228)."""
        sd, p = self.sd, f"xattn_gpt.xattns.{i}."
        E, H = self.embed_dim, self.xattn_heads
        d = E // H
        B, Lq, _ = q.shape
        Lp = kv.shape[1]
        qn = F.layer_norm(q, (E,), sd[p + "layernorm.weight"], sd[p + "layernorm.bias"], 1e-5)
        qs = _lin(qn, sd[p + "query.weight"]).view(B, Lq, H, d).transpose(1, 2)
        k, v = _lin(self._fq("kv", 0, 0, kv), sd[p + "key_value.weight"]).chunk(2, dim=-1)
        k = k.view(B, Lp, H, d).transpose(1, 2)
        v = v.view(B, Lp, H, d).transpose(1, 2)
        s = qs @ k.transpose(-1, -2) / math.sqrt(d)
        s = s + (1.0 - mask[:, None, None, :].float()) * FMIN                       # :197-202,:253-255
        ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, E)
        a = _lin(ctx, sd[p + "attention_out.weight"]) + q                           # :217-218
        f = F.layer_norm(a, (E,), sd[p + "ln.weight"], sd[p + "ln.bias"], 1e-5)
        f = F.gelu(_lin(f, sd[p + "linear1.weight"]))                              # exact erf GELU
        f = f * _lin(a, sd[p + "gated_layer.weight"])                              # gate reads UN-normed a (:224)
        return _lin(f, sd[p + "linear2.weight"]) + a

    def _block(self, i, x, key_mask_add, prefix="xattn_gpt.h."):
        """Block.forward (components.py:23-37), Attention._attn (:51-80), MLP.forward (:97-102); gpt/gpt.py:223-301 is a
        second copy of the same classes (prefix "transformer.lm.h.", baseline policies)."""
        sd, p = self.sd, f"{prefix}{i}."
        E, H = self.embed_dim, self.sattn_heads
        d = E // H
        B, L, _ = x.shape
        qkv = _conv1d(x, sd[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"])
        q, k, v = qkv.split(E, dim=2)
        q = q.view(B, L, H, d).transpose(1, 2)
        k = k.view(B, L, H, d).transpose(1, 2)
        v = v.view(B, L, H, d).transpose(1, 2)
        w = q @ k.transpose(-1, -2) / math.sqrt(d)
        b = torch.tril(torch.ones(L, L, dtype=w.dtype, device=w.device)).view(1, 1, L, L)   # the `attn.bias` buffer
        w = w * b + -1e4 * (1 - b)                                                  # :63
        w = torch.softmax(w + key_mask_add, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(B, L, E)
        a = _conv1d(a, sd[p + "attn.c_proj.weight"], sd[p + "attn.c_proj.bias"])
        n = F.layer_norm(x + a, (E,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        h = F.gelu(_conv1d(n, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        h = h * _lin(n, sd[p + "mlp.gated_layer.weight"])
        m = _conv1d(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        return F.layer_norm(n + m, (E,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)

    def xattn_gpt(self, tokens, pos_ids, prompt, prompt_mask, prompt_pos_ids, masks):
        """XAttnGPT.forward (xattn_gpt.py:73-139), batch-first tensors."""
        sd = self.sd
        n_pos = sd["xattn_gpt.xattn_positions_embed.weight"].shape[0]
        assert prompt.shape[1] <= n_pos                                             # xattn_gpt.py:110
        x = tokens + sd["xattn_gpt.positions_embed.weight"][pos_ids]
        kv = prompt + sd["xattn_gpt.xattn_positions_embed.weight"][prompt_pos_ids]
        key_add = (1.0 - masks[:, None, None, :].float()) * FMIN                    # :116-121
        for i in range(self.n_layers):
            x = self._xattention(i, x, kv, prompt_mask)
            x = self._block(i, x, key_add)
        return x

    def forward(self, obs_token, obs_mask, action_token, prompt_token, prompt_token_mask):
        """VIMAPolicy.forward (vima_policy.py:116-159)."""
        T, B, Q, E = obs_token.shape
        L_act = 0 if action_token is None else action_token.shape[0]
        L = T * Q + L_act
        tokens = torch.zeros(L, B, E, dtype=torch.float32, device=obs_token.device)
        masks = torch.ones(L, B, dtype=torch.bool, device=obs_token.device)
        for t in range(T):
            s = t * (Q + 1)
            tokens[s:s + Q] = obs_token[t].transpose(0, 1)
            masks[s:s + Q] = obs_mask[t].transpose(0, 1)
            if t < L_act:
                tokens[s + Q] = action_token[t]
        pos = (torch.cumsum(masks, dim=0) - 1).long()                               # :145-146
        ppos = torch.cumsum(prompt_token_mask, dim=1) - 1                           # :147
        out = self.xattn_gpt(tokens.transpose(0, 1), pos.transpose(0, 1), prompt_token.transpose(0, 1),
                             prompt_token_mask, ppos, masks.transpose(0, 1))
        out = out.transpose(0, 1)
        return out[Q - 1::Q + 1]                                                    # :158

    __call__ = forward

    def discretize_action(self, action):
        """vima_policy.py:267-299 (bucketize against linspace(0,1,n_bins))."""
        bx = torch.linspace(0, 1, 50)
        by = torch.linspace(0, 1, 100)
        br = torch.linspace(0, 1, 50)
        out = {}
        for k in ("pose0_position", "pose1_position"):
            a = action[k].clone()
            a[..., 0] = torch.bucketize(action[k][..., 0].contiguous(), bx)
            a[..., 1] = torch.bucketize(action[k][..., 1].contiguous(), by)
            out[k] = a.long()
        for k in ("pose0_rotation", "pose1_rotation"):
            out[k] = torch.bucketize(action[k].contiguous(), br).long()
        return out

    # ------------------------------------------------------------------ whole step (bench helper)
    def cold_step(self, prompts, obs, past_actions=None):
        """COLD pass of SURVEY 8(d): prompt assembly + obs tokens + forward + action head -> raw logits [B,700]."""
        ptok, pmask = self.forward_prompt_assembly(prompts)
        otok, omask = self.forward_obs_token(obs)
        atok = None if past_actions is None else self.forward_action_token(past_actions)
        pred = self.forward(otok, omask, atok, ptok, pmask)
        return self.action_logits(pred[-1:])[0]
