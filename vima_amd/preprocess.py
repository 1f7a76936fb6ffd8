"""GPU image preprocessing in front of the policy: the per-object half of the reference's `prepare_obs` /
`prepare_prompt` (/root/reference/scripts/example.py:374-473, :243-371), i.e. segmentation mask -> bbox -> crop ->
pad-to-square -> cv2.resize(32x32, INTER_AREA), as ONE kernel launch per view over all frames and objects
(`vima_crop_objects` in include/vima_hip.h) instead of a numpy + cv2 loop per object on the host. Full frames stay on
the GPU; the outputs are exactly the `cropped_img` / `bbox` / `mask` tensors `VIMAPolicy.forward_obs_token` and
`forward_prompt_assembly` take. No CPU fallback: the HIP library is required."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .containers import MapDict


def _dev_tensor(x, device, dtype=None):
    t = torch.as_tensor(x)
    return t.to(device=device, dtype=dtype if dtype is not None else t.dtype).contiguous()


def crop_objects(rgb, segm, obj_ids, device=None):
    """rgb u8 [..., 3, H, W], segm (uint8 / any integer dtype) [..., H, W], obj_ids: sequence of ints ->
    (crops u8 [..., n_obj, 3, 32, 32], bbox i64 [..., n_obj, 4], mask bool [..., n_obj]) for ONE view, on the GPU."""
    lib = _lib.load()
    rgb = torch.as_tensor(rgb)
    device = torch.device(device) if device is not None else (rgb.device if rgb.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if device.type != "cuda":
        raise RuntimeError("vima_amd.preprocess.crop_objects runs on the GPU only (no CPU fallback)")
    if rgb.dtype != torch.uint8:
        raise AssertionError(f"rgb must be uint8, got {rgb.dtype}")
    rgb = rgb.to(device).contiguous()
    segm = torch.as_tensor(segm)
    if segm.dtype.is_floating_point or segm.dtype == torch.bool:
        raise AssertionError(f"segm must be an integer id map, got {segm.dtype}")
    segm = segm.to(device=device, dtype=torch.uint8 if segm.dtype == torch.uint8 else torch.int32).contiguous()
    lead = tuple(rgb.shape[:-3])
    H, W = int(rgb.shape[-2]), int(rgb.shape[-1])
    if rgb.shape[-3] != 3 or tuple(segm.shape) != lead + (H, W):
        raise ValueError(f"rgb must be [..., 3, H, W] and segm [..., H, W]; got {tuple(rgb.shape)} / {tuple(segm.shape)}")
    ids = torch.as_tensor(list(obj_ids), dtype=torch.int32).to(device)
    n_obj = int(ids.numel())
    n = 1
    for d in lead:
        n *= int(d)
    crops = torch.empty(lead + (n_obj, 3, 32, 32), dtype=torch.uint8, device=device)
    bbox = torch.empty(lead + (n_obj, 4), dtype=torch.int64, device=device)
    mask = torch.empty(lead + (n_obj,), dtype=torch.bool, device=device)
    if n and n_obj:
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(lib.vima_crop_objects(p(rgb), p(segm), segm.element_size(), p(ids), n, n_obj, H, W, p(crops), p(bbox), p(mask),
                                         stream))
    return crops, bbox, mask


def prepare_obs(*, obs: dict, rgb_dict: dict | None = None, meta: dict, device=None):
    """Mirror of the reference's `prepare_obs` (scripts/example.py:374-473), same keyword signature: `obs` holds "ee"
    [L_obs] and "segm"{view} [L_obs,H,W] (and "rgb"{view} [L_obs,3,H,W] unless `rgb_dict` is given); `meta["obj_id_to_info"]`
    lists the object ids. Returns {"ee": [L_obs,1] int64, "objects": {cropped_img, bbox, mask}{view} with leading
    [L_obs, 1, n_obj]} on the GPU -- the structure `VIMAPolicy.forward_obs_token` consumes (batch axis of size 1 like
    the reference's `any_stack([obs_list], dim=0)` + `any_transpose_first_two_axes`). `obs` is consumed ("rgb"/"segm"
    are popped) exactly like the reference does."""
    assert not (rgb_dict is not None and "rgb" in obs)
    rgb_dict = rgb_dict or obs.pop("rgb")
    segm_dict = obs.pop("segm")
    views = sorted(rgb_dict.keys())
    assert meta["n_objects"] == len(meta["obj_id_to_info"])
    objects = list(meta["obj_id_to_info"].keys())
    out = {"cropped_img": {}, "bbox": {}, "mask": {}}
    dev = None
    for view in views:
        crops, bbox, mask = crop_objects(rgb_dict[view], segm_dict[view], objects, device=device)
        dev = crops.device
        out["cropped_img"][view] = crops.unsqueeze(1)
        out["bbox"][view] = bbox.unsqueeze(1)
        out["mask"][view] = mask.unsqueeze(1)
    ee = torch.as_tensor(obs["ee"]).to(device=dev, dtype=torch.int64).reshape(-1, 1)
    return {"ee": ee, "objects": MapDict({k: MapDict(v) for k, v in out.items()})}


def prepare_prompt_images(prompt_assets: dict, placeholders: list[str], views=("front", "top"), device=None):
    """The image half of the reference's `prepare_prompt` (scripts/example.py:256-371; the tokenizer half needs the
    t5-base vocabulary from the Hub): for every placeholder name in prompt order, the objects of that asset
    (`placeholder_type` "object": one id, "scene": every id of `segm["obj_info"]`) are cropped in every view, objects
    that are not visible are DROPPED (not padded: example.py:281-282), and the per-asset lists are right-padded to the
    longest one with zero rows / mask False (example.py:325-352). -> image_batch {cropped_img, bbox, mask}{view} with
    leading [n_placeholders, max_objs] -- the third element of the `prompts` triple of `forward_prompt_assembly`."""
    views = sorted(views)
    per = {v: [] for v in views}
    for name in placeholders:
        asset = prompt_assets[name]
        info = asset["segm"]["obj_info"]
        ids = [info["obj_id"]] if asset["placeholder_type"] == "object" else [e["obj_id"] for e in info]
        for v in views:
            c, b, m = crop_objects(torch.as_tensor(asset["rgb"][v]), torch.as_tensor(asset["segm"][v]), ids, device=device)
            n = int(m.sum().item())                      # present objects are compacted to the front by the kernel
            per[v].append((c[:n], b[:n], m[:n]))
    out = {"cropped_img": {}, "bbox": {}, "mask": {}}
    for v in views:
        mx = max((c.shape[0] for c, _, _ in per[v]), default=0)
        cs, bs, ms = [], [], []
        for c, b, m in per[v]:
            pad = mx - c.shape[0]
            cs.append(torch.cat([c, c.new_zeros(pad, 3, 32, 32)]))
            bs.append(torch.cat([b, b.new_zeros(pad, 4)]))
            ms.append(torch.cat([m, m.new_zeros(pad)]))
        out["cropped_img"][v] = torch.stack(cs) if cs else torch.zeros(0, 0, 3, 32, 32, dtype=torch.uint8)
        out["bbox"][v] = torch.stack(bs) if bs else torch.zeros(0, 0, 4, dtype=torch.int64)
        out["mask"][v] = torch.stack(ms) if ms else torch.zeros(0, 0, dtype=torch.bool)
    return MapDict({k: MapDict(v) for k, v in out.items()})
