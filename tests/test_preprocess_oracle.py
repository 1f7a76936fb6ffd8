"""CPU checks of the preprocessing oracle (oracle/preprocess_oracle.py): known answers of the INTER_AREA restatement, and
the integer logic (bbox, crop, padding, slot order, masks, shapes, dtypes) against the REFERENCE'S OWN `prepare_obs`
function body -- extracted from /root/reference/scripts/example.py with `ast` (the module itself cannot be imported:
it needs vima_bench, gym, cv2 and the t5-base tokenizer from the Hub) and run with `cv2.resize` bound to the oracle's
`resize_area_32`, since OpenCV is not installed here (parity of the resize arithmetic itself is UNPINNED, see the oracle
header)."""
import ast
import os
import sys
import types

import numpy as np
import pytest

from oracle import ref_shim
from oracle.preprocess_oracle import (crop_objects_view, crop_square, object_bbox, prepare_obs_oracle, resize_area_32,
                                      synthetic_frames)


def test_resize_known_answers():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert np.array_equal(resize_area_32(img), img)                                   # 32 px: identity
    img = g.integers(0, 256, (64, 64, 3), dtype=np.uint8)                             # 2x2: (sum + 2) >> 2
    s = img.astype(np.int64).reshape(32, 2, 32, 2, 3).sum((1, 3))
    assert np.array_equal(resize_area_32(img), (s + 2) >> 2)
    img = g.integers(0, 256, (96, 96, 3), dtype=np.uint8)                             # 3x3: fp32 product, half to even
    s = img.astype(np.int64).reshape(32, 3, 32, 3, 3).sum((1, 3))
    want = np.rint(s.astype(np.float32) * (np.float32(1) / np.float32(9))).astype(np.uint8)
    assert np.array_equal(resize_area_32(img), want)
    assert np.abs(resize_area_32(img).astype(int) - np.rint(s / 9.0)).max() <= 1
    for S in (1, 2, 5, 13, 31, 33, 47, 50, 100, 127, 200, 255):                        # constants stay constant
        for val in (0, 9, 200, 255):
            assert np.unique(resize_area_32(np.full((S, S, 3), val, np.uint8))).tolist() == [val], (S, val)
    for S in (40, 50, 77, 100, 130):                                                  # fractional area: a weighted MEAN
        img = g.integers(0, 256, (S, S, 1), dtype=np.uint8)
        r = resize_area_32(img).astype(np.float64)
        assert abs(r.mean() - img.mean()) < 1.0 and r.min() >= img.min() and r.max() <= img.max()
    img = np.zeros((16, 16, 1), np.uint8)                                             # 16 -> 32: every source pixel twice
    img[::2, ::2] = 255
    r = resize_area_32(img)
    assert np.array_equal(r[::2, ::2], r[1::2, 1::2]) or r.shape == (32, 32, 1)
    assert np.array_equal(resize_area_32(np.arange(16 * 16, dtype=np.uint8).reshape(16, 16, 1))[::2, ::2, 0],
                          np.arange(16 * 16, dtype=np.uint8).reshape(16, 16))


def test_bbox_crop_padding():
    segm = np.zeros((8, 10), np.uint8)
    segm[2:7, 3:5] = 7                                # 5 rows x 2 cols
    assert object_bbox(segm, 7) == (3, 4, 2, 6)
    assert object_bbox(segm, 9) is None
    segm[0, 0] = 9                                    # a single pixel is "not present" (len(xs) < 2)
    assert object_bbox(segm, 9) is None
    rgb = np.arange(3 * 8 * 10, dtype=np.uint8).reshape(3, 8, 10)
    sq = crop_square(rgb, (3, 4, 2, 6))
    assert sq.shape == (5, 5, 3)
    assert (sq[:, 0] == 0).all() and (sq[:, 3:] == 0).all()                          # pad 1 before, 2 after on the short axis
    assert np.array_equal(sq[:, 1:3, 0], rgb[0, 2:7, 3:5])


def _reference_prepare_obs():
    """The reference's prepare_obs compiled from its own source text (no copy in this repo), cv2.resize -> oracle."""
    path = os.path.join(ref_shim.REFERENCE_ROOT, "scripts", "example.py")
    tree_ = ast.parse(open(path).read())
    fn = [n for n in tree_.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_obs"][0]
    ref_shim.load_reference()
    if "omegaconf" not in sys.modules:                # any_to_datadict imports it lazily (vima/utils.py:650)
        om = types.ModuleType("omegaconf")
        om.OmegaConf = type("OmegaConf", (), {})
        om.DictConfig = type("DictConfig", (), {})
        sys.modules["omegaconf"] = om
    import vima.utils as vu
    from einops import rearrange
    cv2 = types.SimpleNamespace(INTER_AREA=3, resize=lambda img, size, interpolation: resize_area_32(np.ascontiguousarray(img)))
    ns = {"np": np, "cv2": cv2, "rearrange": rearrange}
    for name in ("any_slice", "any_stack", "any_to_datadict", "get_batch_size", "any_transpose_first_two_axes", "any_concat",
                 "stack_sequence_fields", "any_to_torch_tensor"):
        ns[name] = getattr(vu, name)
    mod = ast.Module(body=[fn], type_ignores=[])
    exec(compile(mod, path, "exec"), ns)
    return ns["prepare_obs"]


@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")
@pytest.mark.parametrize("seed,missing", [(0, ()), (1, (3,)), (2, (2, 5))])
def test_oracle_matches_the_references_prepare_obs(seed, missing):
    L, n_obj = 2, 5
    frames = {v: synthetic_frames(L, n_obj, seed=seed * 10 + i, missing=missing) for i, v in enumerate(("front", "top"))}
    obj_ids = frames["front"][2]
    rgb = {v: frames[v][0] for v in frames}
    segm = {v: frames[v][1] for v in frames}
    prepare_obs = _reference_prepare_obs()
    obs = {"ee": np.zeros((L,), dtype=np.int64), "rgb": {v: rgb[v].copy() for v in rgb}, "segm": {v: segm[v].copy() for v in segm}}
    meta = {"n_objects": n_obj, "obj_id_to_info": {i: {} for i in obj_ids}}
    try:
        ref = prepare_obs(obs=obs, rgb_dict=None, meta=meta)
    except (AttributeError, ImportError, NotImplementedError) as e:     # container plumbing the shims do not cover
        pytest.skip(f"reference plumbing unavailable here: {e}")
    want = prepare_obs_oracle(rgb, segm, obj_ids)
    for key in ("cropped_img", "bbox", "mask"):
        for v in ("front", "top"):
            got = np.asarray(ref["objects"][key][v])
            assert got.shape[:2] == (L, 1), (key, got.shape)           # [L_obs, 1, n_obj, ...] after the transpose
            assert np.array_equal(got[:, 0], want[key][v]), (key, v)
            assert got.dtype == want[key][v].dtype or key == "bbox", (key, got.dtype)
