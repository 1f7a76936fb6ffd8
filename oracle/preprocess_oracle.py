"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of the image preprocessing that sits
immediately in front of the policy -- `prepare_obs` / the image half of `prepare_prompt` in
/root/reference/scripts/example.py:243-473: per (frame, view, object id)

    segmentation mask -> pixel bbox -> crop (inclusive) -> zero-pad to a square -> cv2.resize(32x32, INTER_AREA) -> uint8

PARITY UNPINNED for the resize step: the arithmetic lives in a third-party dependency that is absent from
/root/reference AND from this container -- OpenCV (`import cv2`, scripts/example.py:8; not listed in requirements.txt, no
pinned version; it arrives with vima_bench). `resize_area_32` below restates OpenCV's published algorithm
(modules/imgproc/src/resize.cpp, 4.x: `resizeAreaFast_` / `ResizeAreaFastVec` for integer scale factors,
`computeResizeAreaTab` + `ResizeArea_Invoker` for fractional down-scaling, and the INTER_AREA -> fixed-point bilinear
branch of `cv::resize` with "area" source coordinates when the crop is SMALLER than 32 px) from its documented
behaviour; it could not be checked against cv2 outputs here. What IS pinned: the integer paths (bbox, crop, padding,
slot order, masks) against a literal numpy re-run of the reference's own loop (`reference_loop_obs`, which is the
reference's code path with only the cv2 call swapped for `resize_area_32`), and the resize against known-answer
properties (identity at 32 px, exact block means at integer factors, constants stay constant) and two independent
computations (tests/test_preprocess_crosscheck.py): the geometric definition of area resampling in float64 for the
fractional regime (agreement to half a grey level, i.e. the rounded exact mean) and Pillow's BOX filter for integer
factors (one grey level: Pillow rounds between its passes). That pins the geometry, not OpenCV's rounding.

Everything is integer / uint8 work and is compared BIT-EXACTLY with the HIP kernels (tests/test_preprocess_gpu.py).
"""
from __future__ import annotations

import numpy as np

OUT = 32
COEF_BITS = 11            # INTER_RESIZE_COEF_BITS
COEF_SCALE = 1 << COEF_BITS


# ------------------------------------------------------------------------------------------------ cv2.resize(INTER_AREA)
def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list of (di, si, alpha float32) in emission order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(np.ceil(fsx1))
        sx2 = int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _round_half_even_u8(x):
    """saturate_cast<uchar>(float): cvRound (round half to even) then clamp."""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def _resize_area_fractional(img: np.ndarray) -> np.ndarray:
    """ResizeArea_Invoker<uchar, float>: per source row a horizontally weighted row (fp32, adds in table order), rows
    accumulated as sum = beta*buf (first row of a destination row) / sum += beta*buf, separate multiply and add."""
    S = img.shape[0]
    scale = S / OUT
    tab = _area_tab(S, OUT, scale)                       # same table for x and y (square source)
    out = np.zeros((OUT, OUT, img.shape[2]), dtype=np.uint8)
    src = img.astype(np.float32)
    rows = {}
    for (dy, sy, beta) in tab:
        buf = np.zeros((OUT, img.shape[2]), dtype=np.float32)
        for (dx, sx, alpha) in tab:
            buf[dx] = (buf[dx] + (src[sy, sx] * alpha).astype(np.float32)).astype(np.float32)
        if dy not in rows:
            rows[dy] = (beta * buf).astype(np.float32)
        else:
            rows[dy] = (rows[dy] + (beta * buf).astype(np.float32)).astype(np.float32)
    for dy, s in rows.items():
        out[dy] = _round_half_even_u8(s)
    return out


def _resize_area_integer(img: np.ndarray, f: int) -> np.ndarray:
    """resizeAreaFast_: f x f block sums (int); f == 2 takes ResizeAreaFastVec's (sum + 2) >> 2, every other factor
    saturate_cast<uchar>(sum * (1.f / (f*f))) -- an fp32 product rounded half to even."""
    C = img.shape[2]
    s = img.astype(np.int64).reshape(OUT, f, OUT, f, C).sum(axis=(1, 3))
    if f == 2:
        return ((s + 2) >> 2).astype(np.uint8)
    scale = np.float32(1.0) / np.float32(f * f)
    return _round_half_even_u8((s.astype(np.float32) * scale).astype(np.float32))


def _linear_area_coeffs(ssize: int):
    """x/y tables of cv::resize for INTER_AREA when up-scaling (scale < 1): source index and the fixed-point pair
    (1-f, f) * 2048; the bounds handling of the generic 2-tap path clamps to the last source pixel."""
    scale = ssize / OUT                      # double
    inv = 1.0 / scale
    ofs = np.zeros(OUT, dtype=np.int64)
    co = np.zeros((OUT, 2), dtype=np.int64)
    for d in range(OUT):
        s = int(np.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv)
        f = np.float32(0.0) if f <= 0 else np.float32(f - np.floor(f))
        if s < 0:
            f, s = np.float32(0.0), 0
        if s >= ssize - 1:
            f, s = np.float32(0.0), ssize - 1
        ofs[d] = s
        c0 = np.float32(np.float32(1.0) - f) * np.float32(COEF_SCALE)
        c1 = np.float32(f) * np.float32(COEF_SCALE)
        co[d, 0] = int(np.clip(np.rint(c0), -32768, 32767))      # saturate_cast<short>
        co[d, 1] = int(np.clip(np.rint(c1), -32768, 32767))
    return ofs, co


def _resize_area_upscale(img: np.ndarray) -> np.ndarray:
    """INTER_AREA with a destination LARGER than the source: cv::resize switches to its 8-bit fixed-point bilinear
    (HResizeLinear: int rows S[sx]*a0 + S[sx+1]*a1; VResizeLinear: ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2)
    with the "area" source coordinates sx = floor(dx*scale), f = (dx+1) - (sx+1)/scale."""
    S = img.shape[0]
    ofs, co = _linear_area_coeffs(S)
    src = img.astype(np.int64)
    nxt = np.minimum(ofs + 1, S - 1)
    rows = src[:, ofs, :] * co[None, :, 0, None] + src[:, nxt, :] * co[None, :, 1, None]     # [S, 32, C] horizontal pass
    r0 = rows[ofs]                                                                            # [32, 32, C]
    r1 = rows[nxt]
    b0 = co[:, 0][:, None, None]
    b1 = co[:, 1][:, None, None]
    v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_area_32(img: np.ndarray) -> np.ndarray:
    """cv2.resize(img [S,S,C] uint8, (32, 32), interpolation=cv2.INTER_AREA) for a SQUARE source."""
    S = img.shape[0]
    assert img.shape[1] == S and img.dtype == np.uint8 and S >= 1
    if S == OUT:
        return img.copy()
    if S > OUT:
        if S % OUT == 0:
            return _resize_area_integer(img, S // OUT)
        return _resize_area_fractional(img)
    return _resize_area_upscale(img)


# ------------------------------------------------------------------------------------------------ mask -> bbox -> crop
def object_bbox(segm: np.ndarray, obj_id: int):
    """example.py:400-408: None when the object covers fewer than two pixels, else (xmin, xmax, ymin, ymax)."""
    ys, xs = np.nonzero(segm == obj_id)
    if len(xs) < 2 or len(ys) < 2:
        return None
    return int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())


def crop_square(rgb: np.ndarray, box) -> np.ndarray:
    """example.py:409-423: inclusive crop of rgb [3,H,W], zero-padded to a square (the shorter side is centred, the odd
    pixel of padding goes AFTER) -> [S,S,3]."""
    xmin, xmax, ymin, ymax = box
    c = rgb[:, ymin:ymax + 1, xmin:xmax + 1]
    h, w = c.shape[1], c.shape[2]
    if h != w:
        diff = abs(h - w)
        before, after = diff // 2, diff - diff // 2
        pad = ((0, 0), (0, 0), (before, after)) if h > w else ((0, 0), (before, after), (0, 0))
        c = np.pad(c, pad, mode="constant", constant_values=0)
    return np.ascontiguousarray(np.transpose(c, (1, 2, 0)))


def crop_objects_view(rgb: np.ndarray, segm: np.ndarray, obj_ids):
    """One frame of one view (the body of example.py:395-446): objects that are present come FIRST in `obj_ids` order,
    missing ones are zero rows at the end with mask False. -> crops u8 [n,3,32,32], bbox i64 [n,4] (xc,yc,h,w), mask."""
    n = len(obj_ids)
    crops = np.zeros((n, 3, OUT, OUT), dtype=np.uint8)
    bbox = np.zeros((n, 4), dtype=np.int64)
    mask = np.zeros((n,), dtype=bool)
    slot = 0
    for oid in obj_ids:
        box = object_bbox(segm, oid)
        if box is None:
            continue
        xmin, xmax, ymin, ymax = box
        bbox[slot] = [int((xmin + xmax) / 2), int((ymin + ymax) / 2), ymax - ymin, xmax - xmin]
        crops[slot] = np.transpose(resize_area_32(crop_square(rgb, box)), (2, 0, 1))
        mask[slot] = True
        slot += 1
    return crops, bbox, mask


def prepare_obs_oracle(rgb: dict, segm: dict, obj_ids):
    """prepare_obs (example.py:374-473) without the container plumbing: rgb[view] u8 [L,3,H,W], segm[view] [L,H,W] ->
    {"cropped_img","bbox","mask"}[view] with leading [L, n_obj]."""
    out = {"cropped_img": {}, "bbox": {}, "mask": {}}
    for view in sorted(rgb.keys()):
        cs, bs, ms = [], [], []
        for l in range(rgb[view].shape[0]):
            c, b, m = crop_objects_view(rgb[view][l], segm[view][l], obj_ids)
            cs.append(c), bs.append(b), ms.append(m)
        out["cropped_img"][view] = np.stack(cs)
        out["bbox"][view] = np.stack(bs)
        out["mask"][view] = np.stack(ms)
    return out


# ------------------------------------------------------------------------------------------------ synthetic frames
def synthetic_frames(L: int, n_obj: int, H: int = 128, W: int = 256, seed: int = 0, missing=()):
    """Random rgb + a segmentation made of (possibly overlapping, later objects on top) rectangles and discs of very
    different sizes (1 px to > 128 px: every resize path), object ids 2..n_obj+1 (0/1 = background / table as in
    VIMA-Bench), ids in `missing` never drawn. -> rgb u8 [L,3,H,W], segm u8 [L,H,W], obj_ids."""
    g = np.random.default_rng(seed)
    rgb = g.integers(0, 256, size=(L, 3, H, W), dtype=np.uint8)
    segm = g.integers(0, 2, size=(L, H, W), dtype=np.uint8)
    obj_ids = list(range(2, 2 + n_obj))
    sizes = [1, 2, 3, 5, 8, 13, 16, 21, 31, 32, 33, 40, 47, 64, 65, 90, 96, 100, 127, 128]
    yy, xx = np.mgrid[0:H, 0:W]
    for l in range(L):
        for k, oid in enumerate(obj_ids):
            if oid in missing:
                continue
            hh = int(sizes[int(g.integers(0, len(sizes)))])
            ww = int(sizes[int(g.integers(0, len(sizes)))]) * (2 if g.random() < 0.3 else 1)
            hh, ww = min(hh, H), min(ww, W)
            y0 = int(g.integers(0, H - hh + 1))
            x0 = int(g.integers(0, W - ww + 1))
            if g.random() < 0.5:
                m = (yy >= y0) & (yy < y0 + hh) & (xx >= x0) & (xx < x0 + ww)
            else:
                cy, cx = y0 + (hh - 1) / 2.0, x0 + (ww - 1) / 2.0
                m = ((yy - cy) / max(hh / 2.0, 0.5)) ** 2 + ((xx - cx) / max(ww / 2.0, 0.5)) ** 2 <= 1.0
            segm[l][m] = oid
    return rgb, segm, obj_ids
