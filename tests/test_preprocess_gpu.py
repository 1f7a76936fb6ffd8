"""GPU image preprocessing (vima_crop_objects through the C ABI / vima_amd.preprocess) against the numpy oracle
(oracle/preprocess_oracle.py): uint8 crops, int64 bboxes and masks must be BIT-EXACT, for every resize regime (up-scaling,
identity, integer and fractional area down-scaling), missing objects (slot compaction + padding), both segmentation
dtypes, and through the `prepare_obs` mirror into `VIMAPolicy.forward_obs_token`."""
import numpy as np
import pytest
import torch

from oracle.preprocess_oracle import crop_objects_view, prepare_obs_oracle, synthetic_frames
from vima_amd import preprocess
from vima_testing import synthetic as syn
from tests.gpu_common import loaded_policy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed,n_obj,missing,segm_dtype", [(0, 6, (), np.uint8), (1, 8, (3, 9), np.uint8), (2, 5, (2,), np.int32),
                                                           (3, 12, (), np.int64), (4, 1, (), np.uint8)])
def test_crop_objects_bit_exact(seed, n_obj, missing, segm_dtype):
    L = 6
    rgb, segm, ids = synthetic_frames(L, n_obj, seed=seed, missing=missing)
    crops, bbox, mask = preprocess.crop_objects(torch.from_numpy(rgb), torch.from_numpy(segm.astype(segm_dtype)), ids, device=DEV)
    torch.cuda.synchronize()
    assert crops.shape == (L, n_obj, 3, 32, 32) and crops.dtype == torch.uint8
    assert bbox.shape == (L, n_obj, 4) and bbox.dtype == torch.int64 and mask.dtype == torch.bool
    sides = set()
    for l in range(L):
        c, b, m = crop_objects_view(rgb[l], segm[l], ids)
        assert np.array_equal(mask[l].cpu().numpy(), m), l
        assert np.array_equal(bbox[l].cpu().numpy(), b), l
        assert np.array_equal(crops[l].cpu().numpy(), c), (l, np.abs(crops[l].cpu().numpy().astype(int) - c).max())
        sides |= {int(max(h, w)) + 1 for (_, _, h, w), ok in zip(b, m) if ok}
    assert mask.sum() >= 1


def test_every_resize_regime_is_hit():
    """One frame per crop side S = 1..140 (a centred S x S square object): up-scaling (S < 32), identity (32), 2x2 / 3x3 /
    4x4 integer factors (64, 96, 128) and fractional area weights (everything else)."""
    H, W = 144, 160
    g = np.random.default_rng(7)
    sides = list(range(1, 141))
    rgb = g.integers(0, 256, size=(len(sides), 3, H, W), dtype=np.uint8)
    segm = np.zeros((len(sides), H, W), dtype=np.uint8)
    for i, S in enumerate(sides):
        segm[i, 2:2 + S, 5:5 + S] = 3
        if S == 1:
            segm[i, 2, 6] = 3            # a single pixel would count as "missing": make it 1 x 2 (S = 2 after padding)
    crops, bbox, mask = preprocess.crop_objects(torch.from_numpy(rgb), torch.from_numpy(segm), [3], device=DEV)
    for i, S in enumerate(sides):
        c, b, m = crop_objects_view(rgb[i], segm[i], [3])
        assert bool(mask[i, 0]) and np.array_equal(bbox[i].cpu().numpy(), b)
        got = crops[i].cpu().numpy()
        assert np.array_equal(got, c), (S, np.abs(got.astype(int) - c).max())


def test_rectangular_crops_are_padded_like_the_reference():
    g = np.random.default_rng(11)
    H, W = 128, 256
    rgb = g.integers(0, 256, size=(4, 3, H, W), dtype=np.uint8)
    segm = np.zeros((4, H, W), dtype=np.uint8)
    segm[0, 10:20, 30:31] = 5          # 10 x 1: pad 4 before / 5 after in x
    segm[1, 64:65, 0:256] = 5          # 1 x 256 spans the frame: fractional factor 8 in y-padding
    segm[2, 0:128, 100:103] = 5        # 128 x 3: integer factor 4
    segm[3, 5:50, 7:40] = 5            # 45 x 33
    crops, bbox, mask = preprocess.crop_objects(torch.from_numpy(rgb), torch.from_numpy(segm), [5, 6], device=DEV)
    for i in range(4):
        c, b, m = crop_objects_view(rgb[i], segm[i], [5, 6])
        assert np.array_equal(mask[i].cpu().numpy(), m) and m.tolist() == [True, False]
        assert np.array_equal(bbox[i].cpu().numpy(), b)
        assert np.array_equal(crops[i].cpu().numpy(), c), i


def test_batch_of_bench_size_and_prepare_obs_into_the_policy():
    """256 frames x 2 views x 4 objects (the observation side of BASELINE configs[2]) in one launch per view, spot-checked
    against the oracle, then the `prepare_obs` mirror (scripts/example.py:374-473 signature) feeds forward_obs_token."""
    n, n_obj = 256, 4
    fr = {v: synthetic_frames(n, n_obj, seed=40 + i, missing=(4,) if i else ()) for i, v in enumerate(("front", "top"))}
    ids = fr["front"][2]
    out = {v: preprocess.crop_objects(torch.from_numpy(fr[v][0]).to(DEV), torch.from_numpy(fr[v][1]).to(DEV), ids) for v in fr}
    torch.cuda.synchronize()
    for v in fr:
        for l in (0, 17, 255):
            c, b, m = crop_objects_view(fr[v][0][l], fr[v][1][l], ids)
            assert np.array_equal(out[v][0][l].cpu().numpy(), c) and np.array_equal(out[v][1][l].cpu().numpy(), b)
            assert np.array_equal(out[v][2][l].cpu().numpy(), m)
    # the mirror: L_obs = 3 steps of one episode
    L = 3
    obs = {"ee": np.array([0, 1, 1]), "rgb": {v: fr[v][0][:L] for v in fr}, "segm": {v: fr[v][1][:L] for v in fr}}
    meta = {"n_objects": n_obj, "obj_id_to_info": {i: {} for i in ids}}
    want = prepare_obs_oracle({v: fr[v][0][:L] for v in fr}, {v: fr[v][1][:L] for v in fr}, ids)
    got = preprocess.prepare_obs(obs=obs, rgb_dict=None, meta=meta, device=DEV)
    assert "rgb" not in obs and "segm" not in obs          # consumed like the reference does
    assert got["ee"].shape == (L, 1)
    for key in ("cropped_img", "bbox", "mask"):
        for v in ("front", "top"):
            assert got["objects"][key][v].shape[:3] == (L, 1, n_obj)
            assert np.array_equal(got["objects"][key][v][:, 0].cpu().numpy(), want[key][v]), (key, v)
    cfg = syn.config("2M")
    pol = loaded_policy(cfg, syn.make_state_dict(cfg, 0), "bf16")
    tok, msk = pol.forward_obs_token(got)
    assert tok.shape == (L, 1, 2 * n_obj, cfg.embed_dim) and torch.isfinite(tok).all()
    assert torch.equal(msk[:, 0].cpu(), torch.cat([torch.from_numpy(want["mask"]["front"]), torch.from_numpy(want["mask"]["top"])], dim=-1))


def test_edge_cases_full_frame_object_empty_batch_and_duplicate_ids():
    """An object covering the whole 128 x 256 frame (S = 256: integer factor 8 with 64 rows of zero padding above / below),
    a frame without any listed object (all slots padded), duplicated ids, and an empty batch."""
    g = np.random.default_rng(3)
    H, W = 128, 256
    rgb = g.integers(0, 256, size=(3, 3, H, W), dtype=np.uint8)
    segm = np.zeros((3, H, W), dtype=np.uint8)
    segm[0, :, :] = 9                       # full frame
    segm[1, 3:5, 4:6] = 2                   # only an id that is NOT asked for
    segm[2, 10:40, 10:60] = 9
    ids = [9, 7, 9]                         # id 9 twice: both slots carry the same object (np.nonzero per id in the reference)
    crops, bbox, mask = preprocess.crop_objects(torch.from_numpy(rgb), torch.from_numpy(segm), ids, device=DEV)
    for i in range(3):
        c, b, m = crop_objects_view(rgb[i], segm[i], ids)
        assert np.array_equal(mask[i].cpu().numpy(), m), (i, m)
        assert np.array_equal(bbox[i].cpu().numpy(), b) and np.array_equal(crops[i].cpu().numpy(), c), i
    assert mask[0].tolist() == [True, True, False] and mask[1].tolist() == [False, False, False]
    assert bbox[0, 0].tolist() == [127, 63, 127, 255]
    assert (crops[1] == 0).all()
    e_c, e_b, e_m = preprocess.crop_objects(torch.zeros(0, 3, H, W, dtype=torch.uint8), torch.zeros(0, H, W, dtype=torch.uint8), [1, 2], device=DEV)
    assert e_c.shape == (0, 2, 3, 32, 32) and e_b.shape == (0, 2, 4) and e_m.shape == (0, 2)


def test_argument_errors():
    with pytest.raises(AssertionError):
        preprocess.crop_objects(torch.zeros(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.uint8), [1], device=DEV)     # float rgb
    with pytest.raises(ValueError):
        preprocess.crop_objects(torch.zeros(1, 3, 8, 8, dtype=torch.uint8), torch.zeros(1, 9, 8, dtype=torch.uint8), [1], device=DEV)
    with pytest.raises(RuntimeError):
        preprocess.crop_objects(torch.zeros(1, 3, 8, 400, dtype=torch.uint8), torch.zeros(1, 8, 400, dtype=torch.uint8), [1], device=DEV)
