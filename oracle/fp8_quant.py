"""TEST INFRASTRUCTURE: weight fake-quantisation that mirrors the fp8w packing of the HIP library
(vima_amd/csrc/vima_api.hip: Packer::pack_w / fp8_eligible / f32_to_e4m3) so that the ORACLE can be run on exactly the
weights the fp8w kernels multiply with: every eligible Linear weight is replaced by dequant(quant(w)) with one scale
(amax / 448) per OUTPUT channel of the packed [N, K] matrix and OCP E4M3 round-to-nearest-even. With these weights the
oracle differs from the HIP fp8w path only by bf16 activation rounding (like the bf16 mode), which separates "is the fp8
kernel correct" (tight gate) from "how much accuracy do 3-bit-mantissa weights cost" (reported against the unmodified
reference's goldens)."""
from __future__ import annotations

import re

import torch


def eligible(N: int, K: int) -> bool:
    return K % 64 == 0 and N % 4 == 0 and N * K >= 65536


def _q_rows(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] fp32 -> dequantised fp8 copy, per-row scale."""
    amax = w.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (w / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return q * scale


LINEAR = [  # nn.Linear layout [out, in]: quantised per row
    r"obj_encoder\.cropped_img_encoder\.vit\.blocks\.\d+\.attn\.in_proj_weight$",
    r"obj_encoder\.cropped_img_encoder\.vit\.blocks\.\d+\.attn\.out_proj\.weight$",
    r"obj_encoder\.cropped_img_encoder\.vit\.blocks\.\d+\.mlp\.c_(fc|proj)\.weight$",
    r"obj_encoder\.bbox_mlp\.\w+\.[36]\.weight$",
    r"obj_encoder\.pre_transformer_layer\.\w+\.weight$",
    r"prompt_obj_post_layer\.[036]\.weight$",
    r"t5_prompt_encoder\.t5\.encoder\.block\.\d+\.layer\.0\.SelfAttention\.o\.weight$",
    r"t5_prompt_encoder\.t5\.encoder\.block\.\d+\.layer\.1\.DenseReluDense\.wo\.weight$",
    r"t5_prompt_encoder_post_layer\.weight$",
    r"xattn_gpt\.xattns\.\d+\.(query|key_value|attention_out|linear1|linear2|gated_layer)\.weight$",
    r"xattn_gpt\.h\.\d+\.mlp\.gated_layer\.weight$",
    r"action_encoder\._post_layer\.weight$",
    r"action_decoder\._decoders\.\w+\.mlps\.\d+\.0\.weight$",
]
CONV1D = [  # HF Conv1D [in, out] and `x @ projection`: quantised per column
    r"xattn_gpt\.h\.\d+\.attn\.c_(attn|proj)\.weight$",
    r"xattn_gpt\.h\.\d+\.mlp\.c_(fc|proj)\.weight$",
    r"obj_encoder\.cropped_img_encoder\.vit\.projection$",
]


def fake_quant_state_dict(sd: dict, t5_fused_rms: bool = True) -> dict:
    out = dict(sd)
    for k, w in sd.items():
        if any(re.search(p, k) for p in LINEAR):
            if eligible(w.shape[0], w.shape[1]):
                out[k] = _q_rows(w)
        elif any(re.search(p, k) for p in CONV1D):
            if eligible(w.shape[1], w.shape[0]):
                out[k] = _q_rows(w.t().contiguous()).t().contiguous()
        elif k.endswith("vit.conv1.weight"):
            out[k] = _q_rows(w.reshape(w.shape[0], -1)).reshape(w.shape)
        elif k == "obs_fusion_layer.weight":                       # packed as W[:, :E]; the 2 end-effector columns stay fp32
            E = w.shape[0]
            if eligible(E, E):
                o = w.clone()
                o[:, :E] = _q_rows(w[:, :E].contiguous())
                out[k] = o
    # T5 q / k / v (fused to one [2304, 768] GEMM: per-row scales, so fusing changes nothing) and wi: with the fused
    # RMSNorm the library quantises W diag(g); the oracle multiplies by g itself, so it gets dequant(Q(W g)) / g
    for k, w in sd.items():
        m = re.search(r"(t5_prompt_encoder\.t5\.encoder\.block\.\d+\.layer\.)(0\.SelfAttention\.[qkv]|1\.DenseReluDense\.wi)\.weight$", k)
        if not m:
            continue
        if t5_fused_rms:
            g = sd[m.group(1) + ("0" if "SelfAttention" in k else "1") + ".layer_norm.weight"]
            out[k] = _q_rows(w * g[None, :]) / g[None, :]
        else:
            out[k] = _q_rows(w)
    return out
