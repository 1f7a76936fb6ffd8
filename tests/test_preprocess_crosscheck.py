"""Independent cross-checks of the INTER_AREA restatement in oracle/preprocess_oracle.py (cv2 is absent from this image and from /root/reference, so
the resize stays "parity unpinned" -- DESIGN.md section 6 -- but its arithmetic can be checked against two things that ARE here):
  * the geometric definition of area resampling -- every destination pixel is the coverage-weighted mean of the source pixels under its footprint --
    evaluated in float64 with exact interval overlaps, for the fractional down-scaling regime (what cv::resize's `computeResizeAreaTab` tabulates);
  * Pillow's BOX resampling for INTEGER factors (there a point-sampled box and a coverage-weighted one coincide; Pillow rounds between its two passes).
Both must agree with the oracle to one grey level. This pins the restatement's geometry, not OpenCV's rounding."""
import numpy as np
import pytest

from oracle.preprocess_oracle import resize_area_32


def _area_geometric(img: np.ndarray, d: int = 32) -> np.ndarray:
    s = img.shape[0]
    scale = s / d

    def weights(n_src, n_dst):
        w = np.zeros((n_dst, n_src))
        for o in range(n_dst):
            lo, hi = o * scale, (o + 1) * scale
            for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_src)):
                w[o, i] = max(0.0, min(hi, i + 1) - max(lo, i))
        return w / w.sum(1, keepdims=True)

    w = weights(s, d)
    out = np.einsum("oi,ijc,pj->opc", w, img.astype(np.float64), w)
    return out


@pytest.mark.parametrize("S", [33, 40, 47, 77, 100, 113, 200])
def test_fractional_downscale_matches_the_geometric_definition(S):
    rng = np.random.default_rng(S)
    img = rng.integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    got = resize_area_32(img).astype(np.float64)
    ref = _area_geometric(img)
    assert np.abs(got - ref).max() <= 0.5 + 1e-3, np.abs(got - ref).max()      # the oracle's output is the rounded exact mean


@pytest.mark.parametrize("S", [64, 96, 128, 256])
def test_integer_factors_match_pillow_box(S):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(S)
    img = rng.integers(0, 256, size=(S, S, 3), dtype=np.uint8)
    got = resize_area_32(img).astype(int)
    ref = np.asarray(Image.fromarray(img).resize((32, 32), resample=Image.BOX)).astype(int)
    assert np.abs(got - ref).max() <= 1
