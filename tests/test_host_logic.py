"""Host-side logic of the VIMAPolicy mirror that needs no GPU: prompt assembly index, action (de)discretisation, the
MultiCategorical wrapper, constructor errors, and the algorithmic FLOP count used as the roofline numerator."""
import numpy as np
import pytest
import torch

from oracle.vima_oracle import OraclePolicy
from vima_testing import synthetic as syn
from vima_amd.dists import MultiCategorical
from vima_amd.policy import VIMAPolicy, build_prompt_index


def _loop_index(raw_types, Q):
    """Literal restatement of the reference's assembly order (vima_policy.py:168-233): tokens in order, an image
    contributing its Q object slots (front view objects then top view objects), right-padded to the longest prompt."""
    rows, w, i = [], 0, 0
    for p in raw_types:
        row = []
        for t in p:
            if t == 0:
                row.append(w)
                w += 1
            elif t == 1:
                row.extend(-(i * Q + j) - 2 for j in range(Q))
                i += 1
            else:
                raise ValueError(f"Invalid prompt token type {t}")
        rows.append(row)
    L = max(len(r) for r in rows)
    return np.array([r + [-1] * (L - len(r)) for r in rows], dtype=np.int32), L, w, i


@pytest.mark.parametrize("seed", range(5))
def test_prompt_index_matches_reference_loop_order(seed):
    rng = np.random.default_rng(seed)
    B, Q = int(rng.integers(1, 6)), int(rng.choice([2, 4, 8]))
    raw = [[int(x) for x in rng.integers(0, 2, size=int(rng.integers(1, 12)))] for _ in range(B)]
    src, L, nw, ni = build_prompt_index(raw, Q)
    ref, Lr, nwr, nir = _loop_index(raw, Q)
    assert (L, nw, ni) == (Lr, nwr, nir)
    assert np.array_equal(src[:, :L], ref)


def test_prompt_index_rejects_unknown_token_type():
    with pytest.raises(ValueError):        # vima_policy.py:177
        build_prompt_index([[0, 1, 2]], 4)


def test_prompt_index_words_only_and_empty_rows():
    src, L, nw, ni = build_prompt_index([[0, 0, 0], [0]], 8)
    assert L == 3 and nw == 4 and ni == 0
    assert src.tolist() == [[0, 1, 2], [3, -1, -1]]


def test_action_discretisation_round_trip_matches_oracle():
    cfg = syn.config("2M")
    pol = VIMAPolicy(**cfg.ctor_kwargs())                     # no GPU needed until a native call is made
    orc = OraclePolicy(syn.make_state_dict(cfg, 0), **cfg.ctor_kwargs())
    g = torch.Generator().manual_seed(0)
    act = {"pose0_position": torch.rand(3, 2, 2, generator=g), "pose0_rotation": torch.rand(3, 2, 4, generator=g),
           "pose1_position": torch.rand(3, 2, 2, generator=g), "pose1_rotation": torch.rand(3, 2, 4, generator=g)}
    mine = pol.discretize_action({k: v.clone() for k, v in act.items()})
    ref = orc.discretize_action({k: v.clone() for k, v in act.items()})
    for k in act:
        assert mine[k].dtype == torch.int64 and torch.equal(mine[k], ref[k]), k
    back = pol._de_discretize_actions(mine)
    assert torch.allclose(back["pose0_position"][..., 0], mine["pose0_position"][..., 0].float() / 50)
    assert torch.allclose(back["pose0_position"][..., 1], mine["pose0_position"][..., 1].float() / 100)
    assert torch.allclose(back["pose1_rotation"], mine["pose1_rotation"].float() / 50)


def test_multicategorical_mode_and_normalised_logits():
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 3, 150, generator=g)
    d = MultiCategorical(logits, [50, 100])
    assert torch.equal(d.mode(), torch.stack([logits[..., :50].argmax(-1), logits[..., 50:].argmax(-1)], dim=-1))
    for c, sl in zip(d._dists, (slice(0, 50), slice(50, 150))):          # Categorical normalises: logits - logsumexp
        assert torch.allclose(c.logits, logits[..., sl] - logits[..., sl].logsumexp(-1, keepdim=True), atol=1e-6)


def test_constructor_mirrors_reference_errors_without_gpu():
    with pytest.raises(ValueError):        # components.py:120-123
        VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=7)
    pol = VIMAPolicy(embed_dim=256, xf_n_layers=1, sattn_n_heads=8, xattn_n_heads=8)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):  # no CPU fallback
            pol.forward_obs_token(None)


def test_algorithmic_flops_of_the_headline_config():
    import bench
    cold, warm = bench.flops_per_sample(768, 11, 512, 32 * 8, 8, 1)
    assert abs(cold / 1e9 - 191.74) < 0.05          # 191.74 GFLOP/sample -> 49.09 TFLOP per B=256 step
    assert 0 < warm < cold


def test_bench_helpers_without_gpu():
    """bench.py pieces that do not need the GPU: the committed PMC traffic figure it reports as roofline.traffic, the
    usable-core probe of the CPU baseline and the argument defaults of the driver contract (N=1, K/W that finish fast)."""
    import inspect
    import bench
    t = bench.pmc_traffic()
    assert t is None or t > 1e6                    # bytes per GEMM launch from profiles/r*_pmc_traffic.json
    assert bench.usable_cores() >= 1
    src = inspect.getsource(bench.main)
    for flag, default in (("--gpus", "default=1"), ("--steps", "default=5"), ("--warmup", "default=2")):
        line = next(l for l in src.splitlines() if f'"{flag}"' in l)
        assert default in line, line


def test_obj_inputs_accept_the_reference_datadict():
    """The boundary is fed by the reference's own container in scripts/example.py (`vima.utils.DataDict`,
    /root/reference/vima/utils.py:228, a MutableMapping that is NOT a dict subclass): VIMAPolicy._obj_inputs must flatten
    and order it exactly like the plain-dict path. Needs the reference tree (build container only)."""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present")
    ref_shim.load_reference()
    from vima.utils import DataDict
    cfg = syn.config("2M")
    pol = VIMAPolicy(**cfg.ctor_kwargs())
    obs = syn.make_obs(2, 3, 2, seed=3)
    plain = {k: {v: t for v, t in d.items()} for k, d in obs["objects"].items()}
    dd = DataDict({"objects": plain, "ee": obs["ee"]})
    assert isinstance(dd["objects"], DataDict) and not isinstance(dd["objects"], dict)
    a = pol._obj_inputs(dd["objects"], 2)
    b = pol._obj_inputs(obs["objects"], 2)
    assert a[3:] == b[3:] == (6, 2)
    for xs, ys in zip(a[:3], b[:3]):
        for x, y in zip(xs, ys):
            assert x.dtype == y.dtype and torch.equal(x, y)
    # the prompt side indexes image_batch the same way (vima_policy.py:164,193)
    imgs = syn.make_prompt(2, n_segments=2, words_per_segment=1, q_per_view=2, seed=4)[2]
    dd_img = DataDict({k: {v: t for v, t in d.items()} for k, d in imgs.items()})
    for xs, ys in zip(pol._obj_inputs(dd_img, 1)[:3], pol._obj_inputs(imgs, 1)[:3]):
        for x, y in zip(xs, ys):
            assert torch.equal(x, y)
