"""Seeded synthetic weights and inputs for the VIMA policy hot path.

No checkpoints or datasets exist offline, so parity and
benchmarks are defined on *shared seeded random weights + identical synthetic tensors*.
This module is the single source of both:

  * `make_state_dict(cfg, seed)`  -> dict with exactly the reference's state_dict keys and
    shapes (`vima/__init__.py:11-14` loads it with strict=True).
    Scales follow the reference constructors (0.02-normal GPT weights, width**-0.5 ViT
    parameters, orthogonal-gain MLPs with the 0.01-gain action-head output layer,
    T5 "factor 1.0" scales) but biases / LayerNorm affine terms are made non-trivial so
    that every term of the arithmetic is exercised by the parity tests.
  * `make_prompt(...)`, `make_obs(...)`, `make_actions(...)` -> inputs with the value
    distributions of SURVEY.md section 8(d).

Everything is generated on the CPU with an explicit `torch.Generator`, so the same
(cfg, seed) gives bit-identical tensors here and on the GPU box.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict

import torch

VIEWS = ("front", "top")  # sorted(["front", "top"]) -- obj_encoder.py:31
ACTION_KEYS = ("pose0_position", "pose0_rotation", "pose1_position", "pose1_rotation")
ACTION_DIMS = {  # vima_policy.py:82-87
    "pose0_position": [50, 100],
    "pose0_rotation": [50, 50, 50, 50],
    "pose1_position": [50, 100],
    "pose1_rotation": [50, 50, 50, 50],
}
N_LOGITS = 700
T5_VOCAB = 32128
VIT_WIDTH = 768
VIT_LAYERS = 4
VIT_HEADS = 24
T5_LAYERS = 12
T5_HEADS = 12
T5_DKV = 64
T5_DFF = 3072
T5_BUCKETS = 32
T5_MAXDIST = 128


@dataclass(frozen=True)
class PolicyConfig:
    embed_dim: int
    xf_n_layers: int
    sattn_n_heads: int
    xattn_n_heads: int
    xattn_n_positions: int = 256  # reference hard-codes 256 (vima_policy.py:30)
    n_positions: int = 512        # xattn_gpt.py:18

    def ctor_kwargs(self):
        return dict(embed_dim=self.embed_dim, xf_n_layers=self.xf_n_layers,
                    sattn_n_heads=self.sattn_n_heads, xattn_n_heads=self.xattn_n_heads)

    def asdict(self):
        return asdict(self)


# model-size ladder
CONFIGS = {
    "2M": PolicyConfig(256, 1, 8, 8),
    "4M": PolicyConfig(256, 2, 8, 8),
    "9M": PolicyConfig(320, 3, 10, 10),
    "20M": PolicyConfig(384, 4, 12, 12),
    "43M": PolicyConfig(512, 5, 16, 16),
    "92M": PolicyConfig(640, 7, 20, 20),
    "200M": PolicyConfig(768, 11, 24, 24),
}


def config(name: str, xattn_n_positions: int = 256) -> PolicyConfig:
    c = CONFIGS[name]
    return PolicyConfig(c.embed_dim, c.xf_n_layers, c.sattn_n_heads, c.xattn_n_heads,
                        xattn_n_positions=xattn_n_positions)


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)
        self.sd: dict[str, torch.Tensor] = {}

    def normal(self, key, shape, std, mean=0.0):
        t = torch.empty(*shape, dtype=torch.float32).normal_(mean, std, generator=self.g)
        self.sd[key] = t
        return t

    def const(self, key, t):
        self.sd[key] = t
        return t

    def ln(self, prefix, dim, bias=True):
        self.normal(prefix + ".weight", (dim,), 0.1, 1.0)
        if bias:
            self.normal(prefix + ".bias", (dim,), 0.05)

    def linear(self, prefix, out_f, in_f, std, bias=True, bias_std=0.02):
        self.normal(prefix + ".weight", (out_f, in_f), std)
        if bias:
            self.normal(prefix + ".bias", (out_f,), bias_std)

    def mlp(self, prefix, dims, last_gain=None):
        """build_mlp layout: Sequential indices 0,3,6,... (nn/utils.py:84-91). Orthogonal-like scale."""
        gain = math.sqrt(2.0)
        for i in range(len(dims) - 1):
            in_f, out_f = dims[i], dims[i + 1]
            g = gain
            if last_gain is not None and i == len(dims) - 2:
                g = last_gain
            self.linear(f"{prefix}.{3 * i}", out_f, in_f, g / math.sqrt(max(in_f, out_f)))


def make_state_dict(cfg: PolicyConfig, seed: int = 0, head_gain: float = 0.01) -> dict[str, torch.Tensor]:
    """State dict with the reference `VIMAPolicy` key layout. `head_gain` is the orthogonal gain
    of the action head's output layers (reference: 0.01, action_decoder.py:150-153 -> logits O(0.06)); the parity
    suite also runs a gain-0.5 variant whose logits are O(1) so that argmax agreement is a meaningful check."""
    E, N = cfg.embed_dim, cfg.xf_n_layers
    I = _Init(seed)
    # ---- xattn_gpt (xattn_gpt.py:45-68, components.py) ----
    p = "xattn_gpt."
    I.const(p + "position_ids", torch.arange(cfg.n_positions))
    I.const(p + "xattn_position_ids", torch.arange(cfg.xattn_n_positions))
    I.normal(p + "positions_embed.weight", (cfg.n_positions, E), 0.02)
    I.normal(p + "xattn_positions_embed.weight", (cfg.xattn_n_positions, E), 0.02)
    tril = torch.tril(torch.ones(cfg.n_positions, cfg.n_positions)).view(1, 1, cfg.n_positions, cfg.n_positions)
    for i in range(N):
        h = f"{p}h.{i}."
        I.const(h + "attn.bias", tril)
        w = I.normal(h + "attn.c_attn.weight", (E, 3 * E), 0.02)   # Conv1D: [in, out]
        w[:, : 2 * E] *= 3.0                                        # peaked q.k logits
        I.normal(h + "attn.c_attn.bias", (3 * E,), 0.02)
        I.normal(h + "attn.c_proj.weight", (E, E), 0.02)
        I.normal(h + "attn.c_proj.bias", (E,), 0.02)
        I.ln(h + "ln_1", E)
        I.normal(h + "mlp.c_fc.weight", (E, 4 * E), 0.02)
        I.normal(h + "mlp.c_fc.bias", (4 * E,), 0.02)
        I.normal(h + "mlp.c_proj.weight", (4 * E, E), 0.02)
        I.normal(h + "mlp.c_proj.bias", (E,), 0.02)
        I.normal(h + "mlp.gated_layer.weight", (4 * E, E), 0.02)
        I.ln(h + "ln_2", E)
    for i in range(N):
        x = f"{p}xattns.{i}."
        I.const(x + "kv_position_ids", torch.arange(cfg.xattn_n_positions))
        I.ln(x + "layernorm", E)
        I.normal(x + "query.weight", (E, E), 0.06)
        kv = I.normal(x + "key_value.weight", (2 * E, E), 0.02)
        kv[:E] *= 3.0
        I.normal(x + "attention_out.weight", (E, E), 0.02)
        I.ln(x + "ln", E)
        I.normal(x + "linear1.weight", (4 * E, E), 0.02)
        I.normal(x + "linear2.weight", (E, 4 * E), 0.02)
        I.normal(x + "gated_layer.weight", (4 * E, E), 0.02)
    # ---- obj_encoder (obj_encoder.py:15-64, vit.py:137-169) ----
    v = "obj_encoder.cropped_img_encoder.vit."
    W = VIT_WIDTH
    sc = W ** -0.5
    I.normal(v + "cls_token", (W,), sc)
    I.normal(v + "pos_embed", (5, W), sc)
    I.normal(v + "projection", (W, W), sc)
    I.normal(v + "conv1.weight", (W, 3, 16, 16), 0.02)
    I.ln(v + "ln_pre", W)
    for j in range(VIT_LAYERS):
        b = f"{v}blocks.{j}."
        w = I.normal(b + "attn.in_proj_weight", (3 * W, W), 0.0255)
        w[: 2 * W] *= 2.5
        I.normal(b + "attn.in_proj_bias", (3 * W,), 0.02)
        I.linear(b + "attn.out_proj", W, W, 0.0209)
        I.ln(b + "ln_1", W)
        I.linear(b + "mlp.c_fc", 4 * W, W, 0.0208)
        I.linear(b + "mlp.c_proj", W, 4 * W, 0.0104)
        I.ln(b + "ln_2", W)
    I.ln(v + "ln_post", W)
    for view in VIEWS:
        I.mlp(f"obj_encoder.bbox_mlp.{view}", [4, 768, 768, 768])
    for view in VIEWS:
        I.linear(f"obj_encoder.pre_transformer_layer.{view}", E, 2 * W, (2 * W) ** -0.5 * 0.58)
    # ---- obs fusion (vima_policy.py:47-49) ----
    I.normal("end_effector_encoder.weight", (2, 2), 0.7)
    I.linear("obs_fusion_layer", E, E + 2, (E + 2) ** -0.5 * 0.58)
    # ---- action encoder (vima_policy.py:51-80, action_embd.py) ----
    for k in ACTION_KEYS:
        in_dim = 2 if k.endswith("position") else 4
        I.mlp(f"action_encoder._embed_dict.{k}._layer", [in_dim, 256, 256])
    if E != 1024:   # nn.Identity when embed_dim == 4 * 256 (action_embd.py:16-20)
        I.linear("action_encoder._post_layer", E, 1024, 1024 ** -0.5 * 0.58)
    # ---- action decoder (action_decoder.py:128-166) ----
    for k in ACTION_KEYS:
        for j, bins in enumerate(ACTION_DIMS[k]):
            I.mlp(f"action_decoder._decoders.{k}.mlps.{j}", [E, 512, 512, bins], last_gain=head_gain)
    # ---- word embedding + T5 encoder (word_embd.py, prompt_encoder.py) ----
    I.normal("prompt_embedding._embed_layer.weight", (T5_VOCAB, 768), 1.0)
    t5 = "t5_prompt_encoder.t5."
    dead = torch.zeros(T5_VOCAB, 768)  # never read at inference (inputs_embeds path); strict load needs the keys
    I.const(t5 + "shared.weight", dead)
    I.const(t5 + "encoder.embed_tokens.weight", dead)
    d_model, inner = 768, T5_HEADS * T5_DKV
    for l in range(T5_LAYERS):
        a = f"{t5}encoder.block.{l}.layer.0."
        I.normal(a + "SelfAttention.q.weight", (inner, d_model), (d_model * T5_DKV) ** -0.5 * 2.0)
        I.normal(a + "SelfAttention.k.weight", (inner, d_model), d_model ** -0.5 * 1.5)
        I.normal(a + "SelfAttention.v.weight", (inner, d_model), d_model ** -0.5)
        I.normal(a + "SelfAttention.o.weight", (d_model, inner), inner ** -0.5)
        if l == 0:
            I.normal(a + "SelfAttention.relative_attention_bias.weight", (T5_BUCKETS, T5_HEADS), 1.0)
        I.normal(a + "layer_norm.weight", (d_model,), 0.1, 1.0)
        f = f"{t5}encoder.block.{l}.layer.1."
        I.normal(f + "DenseReluDense.wi.weight", (T5_DFF, d_model), d_model ** -0.5)
        I.normal(f + "DenseReluDense.wo.weight", (d_model, T5_DFF), T5_DFF ** -0.5)
        I.normal(f + "layer_norm.weight", (d_model,), 0.1, 1.0)
    I.normal(t5 + "encoder.final_layer_norm.weight", (d_model,), 0.1, 1.0)
    if E != 768:
        I.normal("t5_prompt_encoder_post_layer.weight", (E, 768), 768 ** -0.5)
    I.mlp("prompt_obj_post_layer", [E, 768, 768, 768])
    return I.sd


def state_dict_checksum(sd) -> float:
    """Cheap fingerprint used by the golden fixtures to detect RNG drift."""
    tot = 0.0
    for k in sorted(sd):
        t = sd[k]
        if t.dtype.is_floating_point and t.numel() < 5_000_000:
            tot += float(t.double().abs().sum())
    return tot


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------
from vima_amd.containers import MapDict  # noqa: E402,F401  (the product container; re-exported for the tests)


def _objects(g, lead, q_per_view, mask_frac=0.1):
    """crops uint8 U{0..255}; bbox (xc,yc,h,w) ranges of SURVEY 8(d); masks ~mask_frac False,
    never the first object of the front view (else position id -1, vima_policy.py:145)."""
    crops, bbox, mask = {}, {}, {}
    for vi, view in enumerate(VIEWS):
        crops[view] = torch.randint(0, 256, (*lead, q_per_view, 3, 32, 32), generator=g, dtype=torch.int64).to(torch.uint8)
        xc = torch.randint(0, 256, (*lead, q_per_view, 1), generator=g)
        yc = torch.randint(0, 128, (*lead, q_per_view, 1), generator=g)
        hh = torch.randint(1, 128, (*lead, q_per_view, 1), generator=g)
        ww = torch.randint(1, 256, (*lead, q_per_view, 1), generator=g)
        bbox[view] = torch.cat([xc, yc, hh, ww], dim=-1)
        m = torch.rand(*lead, q_per_view, generator=g) >= mask_frac
        if vi == 0:
            m[..., 0] = True
        mask[view] = m
    return MapDict(cropped_img=MapDict(crops), bbox=MapDict(bbox), mask=MapDict(mask))


def make_prompt(batch: int, layout: list[list[int]] | None = None, *, n_segments: int = 2,
                words_per_segment: int = 4, q_per_view: int = 2, seed: int = 1234, mask_frac: float = 0.1):
    """Returns the `prompts` triple consumed by `forward_prompt_assembly` (vima_policy.py:161-162):
    (raw_prompts_token_type, word_batch, image_batch). Default layout: `n_segments` x
    [words_per_segment words, 1 image -> 2*q_per_view object tokens]."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if layout is None:
        layout = [([0] * words_per_segment + [1]) * n_segments for _ in range(batch)]
    assert len(layout) == batch
    n_words = sum(t == 0 for p in layout for t in p)
    n_img = sum(t == 1 for p in layout for t in p)
    word_batch = torch.randint(0, 32100, (n_words,), generator=g)
    image_batch = _objects(g, (max(n_img, 0),), q_per_view, mask_frac)
    return layout, word_batch, image_batch


def cut_prompt(prompts, idx):
    """Sub-batch `idx` (sample indices) of a `make_prompt` triple: words and images are packed per sample in prompt
    order, so sample s owns a contiguous run of each."""
    layout, words, imgs = prompts
    nw = [sum(t == 0 for t in p) for p in layout]
    ni = [sum(t == 1 for t in p) for p in layout]
    w0 = [0]
    i0 = [0]
    for a, b in zip(nw, ni):
        w0.append(w0[-1] + a)
        i0.append(i0[-1] + b)
    wsel = torch.cat([words[w0[s]:w0[s + 1]] for s in idx]) if len(idx) else words[:0]
    isel = MapDict({k: MapDict({v: torch.cat([imgs[k][v][i0[s]:i0[s + 1]] for s in idx]) for v in imgs[k]}) for k in imgs})
    return [layout[s] for s in idx], wsel, isel


def concat_prompts(*parts):
    """Batch-concatenation of `make_prompt` triples (the inverse of `cut_prompt`): samples of part 0 first, then part 1, ...
    Words and images stay packed per sample in prompt order, so every sample keeps exactly the tensors it had alone."""
    layout = [p for part in parts for p in part[0]]
    words = torch.cat([part[1] for part in parts])
    imgs0 = parts[0][2]
    imgs = MapDict({k: MapDict({v: torch.cat([part[2][k][v] for part in parts]) for v in imgs0[k]}) for k in imgs0})
    return layout, words, imgs


def concat_obs(*parts):
    """Batch-concatenation (dim 1) of `make_obs` dicts."""
    o0 = parts[0]["objects"]
    return {"objects": MapDict({k: MapDict({v: torch.cat([p["objects"][k][v] for p in parts], dim=1) for v in o0[k]}) for k in o0}),
            "ee": torch.cat([p["ee"] for p in parts], dim=1)}


def cut_obs(obs, idx):
    """Sub-batch `idx` of a `make_obs` dict (batch is dim 1)."""
    return {"objects": MapDict({k: MapDict({v: obs["objects"][k][v][:, idx] for v in obs["objects"][k]})
                                for k in obs["objects"]}), "ee": obs["ee"][:, idx]}


def cut_actions(actions, idx):
    return None if actions is None else {k: v[:, idx] for k, v in actions.items()}


def prompt_len(layout, q_per_view):
    return max(sum(1 if t == 0 else 2 * q_per_view for t in p) for p in layout)


def make_obs(steps: int, batch: int, q_per_view: int, seed: int = 4321, mask_frac: float = 0.1):
    """obs dict consumed by `forward_obs_token` (vima_policy.py:242-259)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    objects = _objects(g, (steps, batch), q_per_view, mask_frac)
    ee = torch.randint(0, 2, (steps, batch), generator=g)
    return {"objects": objects, "ee": ee}


def make_actions(steps: int, batch: int, seed: int = 777):
    """Discrete past actions (int64 bin indices) as produced by `MultiCategorical.mode()`."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = {}
    for k in ACTION_KEYS:
        cols = [torch.randint(0, b, (steps, batch, 1), generator=g) for b in ACTION_DIMS[k]]
        out[k] = torch.cat(cols, dim=-1)
    return out


def to_device(x, device):
    if isinstance(x, dict):
        return type(x)({k: to_device(v, device) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(to_device(v, device) for v in x)
    if torch.is_tensor(x):
        return x.to(device)
    return x


# --------------------------------------------------------------------------------------
# baseline policies: VIMAGPTPolicy / VIMAGatoPolicy / VIMAFlamingoPolicy
# --------------------------------------------------------------------------------------
RGB_H, RGB_W, RGB_PATCH = 64, 128, 32      # img_size=(64, 128), vit_patch_size=32 (vima_gpt_policy.py:37-45)
RGB_PATCHES = (RGB_H // RGB_PATCH) * (RGB_W // RGB_PATCH)
PERCEIVER_LATENTS, PERCEIVER_BLOCKS, PERCEIVER_SELF_PER_BLOCK, PERCEIVER_HEADS = 4, 4, 4, 8   # vima_flamingo_policy.py:40-45


@dataclass(frozen=True)
class BaselineConfig:
    """Constructor arguments of the reference's baseline policies. kind "gpt" / "gato": decoder-only HFGPT over
    [prompt | sep | obs / action tokens] (vima_gpt_policy.py:12-20, vima_gato_policy.py:13-21); "flamingo": XAttnGPT over
    Perceiver-resampled image tokens (vima_flamingo_policy.py:11-18; xattn_n_positions is hard-coded 256 there)."""
    kind: str
    embed_dim: int
    n_layer: int
    n_head: int
    xattn_n_heads: int = 0
    vocab_size: int = 40478
    n_positions: int = 512
    xattn_n_positions: int = 256

    def ctor_kwargs(self):
        if self.kind == "flamingo":
            return dict(embed_dim=self.embed_dim, dt_n_layers=self.n_layer, dt_n_heads=self.n_head, xattn_n_heads=self.xattn_n_heads)
        return dict(embed_dim=self.embed_dim, vocab_size=self.vocab_size, n_positions=self.n_positions, n_layer=self.n_layer,
                    n_head=self.n_head)

    @property
    def obs_tokens(self):
        """tokens one observation contributes to the sequence: 1 (gpt: cls features of both views concatenated),
        2 * 8 patch tokens (gato), 4 latents (flamingo)"""
        return {"gpt": 1, "gato": 2 * RGB_PATCHES, "flamingo": PERCEIVER_LATENTS}[self.kind]

    @property
    def obj_dim(self):
        """obj_encoder.output_dim (obj_encoder.py:243-245: emb_dim * n_views for MultiViewRGBEncoder)"""
        return 2 * self.embed_dim if self.kind == "gpt" else self.embed_dim

    def asdict(self):
        return asdict(self)


def _gpt_block(I, h, E, n_positions, with_bias_buffer):
    if with_bias_buffer:   # components.py registers the causal mask as a persistent buffer; gpt/gpt.py inherits HF's non-persistent one
        I.const(h + "attn.bias", torch.tril(torch.ones(n_positions, n_positions)).view(1, 1, n_positions, n_positions))
    w = I.normal(h + "attn.c_attn.weight", (E, 3 * E), 0.02)
    w[:, : 2 * E] *= 3.0
    I.normal(h + "attn.c_attn.bias", (3 * E,), 0.02)
    I.normal(h + "attn.c_proj.weight", (E, E), 0.02)
    I.normal(h + "attn.c_proj.bias", (E,), 0.02)
    I.ln(h + "ln_1", E)
    I.normal(h + "mlp.c_fc.weight", (E, 4 * E), 0.02)
    I.normal(h + "mlp.c_fc.bias", (4 * E,), 0.02)
    I.normal(h + "mlp.c_proj.weight", (4 * E, E), 0.02)
    I.normal(h + "mlp.c_proj.bias", (E,), 0.02)
    I.normal(h + "mlp.gated_layer.weight", (4 * E, E), 0.02)
    I.ln(h + "ln_2", E)


def make_baseline_state_dict(cfg: BaselineConfig, seed: int = 0, head_gain: float = 0.01) -> dict[str, torch.Tensor]:
    """State dict with the key layout of the reference's baseline policies (probed from the reference modules in the
    build container; tests/test_baseline_oracle.py loads it into them with strict=True)."""
    E, N = cfg.embed_dim, cfg.n_layer
    I = _Init(seed)
    if cfg.kind == "flamingo":
        p = "xattn_gpt."
        I.const(p + "position_ids", torch.arange(cfg.n_positions))
        I.const(p + "xattn_position_ids", torch.arange(cfg.xattn_n_positions))
        I.normal(p + "positions_embed.weight", (cfg.n_positions, E), 0.02)
        I.normal(p + "xattn_positions_embed.weight", (cfg.xattn_n_positions, E), 0.02)
        for i in range(N):
            _gpt_block(I, f"{p}h.{i}.", E, cfg.n_positions, True)
        for i in range(N):
            x = f"{p}xattns.{i}."
            I.const(x + "kv_position_ids", torch.arange(cfg.xattn_n_positions))
            I.ln(x + "layernorm", E)
            I.normal(x + "query.weight", (E, E), 0.06)
            kv = I.normal(x + "key_value.weight", (2 * E, E), 0.02)
            kv[:E] *= 3.0
            I.normal(x + "attention_out.weight", (E, E), 0.02)
            I.ln(x + "ln", E)
            I.normal(x + "linear1.weight", (4 * E, E), 0.02)
            I.normal(x + "linear2.weight", (E, 4 * E), 0.02)
            I.normal(x + "gated_layer.weight", (4 * E, E), 0.02)
    else:
        I.normal("prompt_sep_token", (E,), 0.5)
        p = "transformer.lm."
        I.const(p + "position_ids", torch.arange(cfg.n_positions))
        I.normal(p + "tokens_embed.weight", (cfg.vocab_size, E), 0.02)     # never read (inputs_embeds path, gpt.py:69-73)
        I.normal(p + "positions_embed.weight", (cfg.n_positions, E), 0.02)
        for i in range(N):
            _gpt_block(I, f"{p}h.{i}.", E, cfg.n_positions, False)
    # ---- image encoder: (Gato)VisionTransformerRectangular (vit.py:83-135, :262-329) on 64x128 frames, 32x32 patches ----
    v = "obj_encoder.cropped_img_encoder.vit."
    W = VIT_WIDTH
    sc = W ** -0.5
    if cfg.kind == "gpt":
        I.normal(v + "cls_token", (W,), sc)
    I.normal(v + "pos_embed", (RGB_PATCHES + (1 if cfg.kind == "gpt" else 0), W), sc)
    I.normal(v + "projection", (W, E), sc)
    I.normal(v + "conv1.weight", (W, 3, RGB_PATCH, RGB_PATCH), 0.01)
    I.ln(v + "ln_pre", W)
    for j in range(VIT_LAYERS):
        b = f"{v}blocks.{j}."
        w = I.normal(b + "attn.in_proj_weight", (3 * W, W), 0.0255)
        w[: 2 * W] *= 2.5
        I.normal(b + "attn.in_proj_bias", (3 * W,), 0.02)
        I.linear(b + "attn.out_proj", W, W, 0.0209)
        I.ln(b + "ln_1", W)
        I.linear(b + "mlp.c_fc", 4 * W, W, 0.0208)
        I.linear(b + "mlp.c_proj", W, 4 * W, 0.0104)
        I.ln(b + "ln_2", W)
    I.ln(v + "ln_post", W)
    if cfg.kind == "flamingo":   # HF PerceiverModel (modeling_perceiver.py:125-527), d_model = d_latents = E
        pc = "obj_encoder.peceiver.model."     # (sic: the reference attribute is spelled `peceiver`, obj_encoder.py:177)
        I.normal(pc + "embeddings.latents", (PERCEIVER_LATENTS, E), 1.0)

        def layer(pre, cross):
            a = pre + "attention."
            I.ln(a + "self.layernorm1", E)
            if cross:
                I.ln(a + "self.layernorm2", E)
            I.linear(a + "self.query", E, E, E ** -0.5 * 1.5)
            I.linear(a + "self.key", E, E, E ** -0.5 * 1.5)
            I.linear(a + "self.value", E, E, E ** -0.5)
            I.linear(a + "output.dense", E, E, E ** -0.5 * 0.5)
            I.ln(pre + "layernorm", E)
            I.linear(pre + "mlp.dense1", E, E, E ** -0.5)     # widening factor 1 (PerceiverConfig defaults)
            I.linear(pre + "mlp.dense2", E, E, E ** -0.5 * 0.5)
        layer(pc + "encoder.cross_attention.", True)
        for i in range(PERCEIVER_SELF_PER_BLOCK):
            layer(f"{pc}encoder.self_attends.{i}.", False)
    # ---- obs fusion / action encoder / action decoder: same modules as VIMAPolicy ----
    I.normal("end_effector_encoder.weight", (2, 2), 0.7)
    I.linear("obs_fusion_layer", E, cfg.obj_dim + 2, (cfg.obj_dim + 2) ** -0.5 * 0.58)
    for k in ACTION_KEYS:
        in_dim = 2 if k.endswith("position") else 4
        I.mlp(f"action_encoder._embed_dict.{k}._layer", [in_dim, 256, 256])
    if E != 1024:
        I.linear("action_encoder._post_layer", E, 1024, 1024 ** -0.5 * 0.58)
    for k in ACTION_KEYS:
        for j, bins in enumerate(ACTION_DIMS[k]):
            I.mlp(f"action_decoder._decoders.{k}.mlps.{j}", [E, 512, 512, bins], last_gain=head_gain)
    # ---- word embedding + T5 encoder ----
    I.normal("prompt_embedding._embed_layer.weight", (T5_VOCAB, 768), 1.0)
    t5 = "t5_prompt_encoder.t5."
    dead = torch.zeros(T5_VOCAB, 768)
    I.const(t5 + "shared.weight", dead)
    I.const(t5 + "encoder.embed_tokens.weight", dead)
    d_model, inner = 768, T5_HEADS * T5_DKV
    for l in range(T5_LAYERS):
        a = f"{t5}encoder.block.{l}.layer.0."
        I.normal(a + "SelfAttention.q.weight", (inner, d_model), (d_model * T5_DKV) ** -0.5 * 2.0)
        I.normal(a + "SelfAttention.k.weight", (inner, d_model), d_model ** -0.5 * 1.5)
        I.normal(a + "SelfAttention.v.weight", (inner, d_model), d_model ** -0.5)
        I.normal(a + "SelfAttention.o.weight", (d_model, inner), inner ** -0.5)
        if l == 0:
            I.normal(a + "SelfAttention.relative_attention_bias.weight", (T5_BUCKETS, T5_HEADS), 1.0)
        I.normal(a + "layer_norm.weight", (d_model,), 0.1, 1.0)
        f = f"{t5}encoder.block.{l}.layer.1."
        I.normal(f + "DenseReluDense.wi.weight", (T5_DFF, d_model), d_model ** -0.5)
        I.normal(f + "DenseReluDense.wo.weight", (d_model, T5_DFF), T5_DFF ** -0.5)
        I.normal(f + "layer_norm.weight", (d_model,), 0.1, 1.0)
    I.normal(t5 + "encoder.final_layer_norm.weight", (d_model,), 0.1, 1.0)
    if E != 768:
        I.normal("t5_prompt_encoder_post_layer.weight", (E, 768), 768 ** -0.5)
    I.mlp("prompt_obj_post_layer", [cfg.obj_dim, 768, 768, 768])
    return I.sd


def make_rgb_prompt(batch: int, layout: list[list[int]] | None = None, *, n_segments: int = 2, words_per_segment: int = 4,
                    seed: int = 1234):
    """`prompts` triple of the baseline policies (vima_gpt_policy.py:189-190): image_batch = {"rgb": {view: u8 [n_img,3,64,128]}}
    (whole frames instead of object crops)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if layout is None:
        layout = [([0] * words_per_segment + [1]) * n_segments for _ in range(batch)]
    assert len(layout) == batch
    n_words = sum(t == 0 for p in layout for t in p)
    n_img = sum(t == 1 for p in layout for t in p)
    word_batch = torch.randint(0, 32100, (n_words,), generator=g)
    rgb = {v: torch.randint(0, 256, (n_img, 3, RGB_H, RGB_W), generator=g, dtype=torch.int64).to(torch.uint8) for v in VIEWS}
    return layout, word_batch, MapDict(rgb=MapDict(rgb))


def make_rgb_obs(steps: int, batch: int, seed: int = 4321):
    """obs dict of the baseline policies (vima_gpt_policy.py:249-259): {"rgb": {view: u8 [T,B,3,64,128]}, "ee": i64 [T,B]}"""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    rgb = {v: torch.randint(0, 256, (steps, batch, 3, RGB_H, RGB_W), generator=g, dtype=torch.int64).to(torch.uint8) for v in VIEWS}
    ee = torch.randint(0, 2, (steps, batch), generator=g)
    return MapDict(rgb=MapDict(rgb), ee=ee)
