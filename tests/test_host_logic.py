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


def _pmc_db(path, counter, rows):
    """a rocprofv3-like rocpd database reduced to what scripts/pmc_summary.py reads: counters_collection(dispatch_id, kernel_name, counter_name, value)"""
    import sqlite3
    c = sqlite3.connect(path)
    c.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real)")
    for i, (k, v) in enumerate(rows):
        for part in (0.5 * v, 1.5 * v):                     # several rows per dispatch (one per XCD in the real tool): the dispatch's value is their mean
            c.execute("insert into counters_collection values (?, ?, ?, ?)", (i + 1, k, counter, part))
    c.commit()
    c.close()


def test_pmc_summary_attributes_traffic_to_gemm_shapes_by_dispatch_order(tmp_path):
    """bench.py's live `roofline.traffic_per_shape` (scripts/pmc_summary.py): per-dispatch FETCH_SIZE / WRITE_SIZE values of two separate passes are joined by
    position with the GEMM launch log of one step; read bytes carry the gfx950 half-count correction (2 x FETCH_SIZE KiB); any disagreement between the trace
    and the log voids the attribution instead of guessing."""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(os.path.dirname(__file__), "..", "scripts", "pmc_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    pp1 = "void vima::(anonymous namespace)::gemm_pp_kernel<0, 1, false>(vima::(anonymous namespace)::GemmDev)"
    pp4 = "void vima::(anonymous namespace)::gemm_pp_kernel<0, 4, false>(vima::(anonymous namespace)::GemmDev)"
    ln = "void vima::(anonymous namespace)::layernorm_rows2_kernel<3>(float const*)"
    step = [(pp1, 1000.0), (ln, 10.0), (pp4, 3000.0), (pp1, 2000.0)]           # KiB per dispatch; the non-GEMM kernel is skipped by the join
    fetch, write = str(tmp_path / "f.db"), str(tmp_path / "w.db")
    _pmc_db(fetch, "FETCH_SIZE", step + step)                                    # two steps in the trace
    _pmc_db(write, "WRITE_SIZE", [(k, v / 2) for k, v in step + step])
    log = [{"kernel": "vima::gemm_pp_kernel<0, 1, false>", "M": 256, "N": 2304, "K": 768},
           {"kernel": "vima::gemm_pp_kernel<0, 4, false>", "M": 256, "N": 768, "K": 3072},
           {"kernel": "vima::gemm_pp_kernel<0, 1, false>", "M": 256, "N": 1536, "K": 768}]
    lp = str(tmp_path / "log.json")
    json.dump(log, open(lp, "w"))
    s = ps.summarise(fetch, write, lp)
    assert s["per_shape"] is not None, s["per_shape_note"]
    by = {(r["M"], r["N"], r["K"]): r for r in s["per_shape"]}
    assert set(by) == {(256, 2304, 768), (256, 768, 3072), (256, 1536, 768)}
    r = by[(256, 768, 3072)]
    assert r["launches_per_step"] == 1 and r["kernel"].startswith("vima::gemm_pp_kernel<0, 4")
    assert abs(r["fetch_mb"] - 2 * 3000.0 * 1024 / 1e6) < 1e-9 and abs(r["write_mb"] - 1500.0 * 1024 / 1e6) < 1e-9
    assert abs(r["bytes_per_launch"] - (2 * 3000.0 + 1500.0) * 1024) < 1e-6
    assert abs(by[(256, 1536, 768)]["bytes_per_launch"] - (2 * 2000.0 + 1000.0) * 1024) < 1e-6          # same kernel, different shape: kept apart
    assert s["gemm_bf16_launches"] == 6
    assert abs(s["gemm_bf16_bytes_per_launch"] - (2 * 2000.0 + 1000.0) * 1024) < 1e-6                    # mean over the six GEMM dispatches
    # the two spellings of one kernel: rocprofv3's full template arguments against the library's launch-log names (VIMAPolicy._gemm_kernel_name)
    V = "void vima::(anonymous namespace)::"
    same = [("vima::gemm_kernel<Tile<64, 64>> act 0", V + "gemm_kernel<unsigned short, vima::(anonymous namespace)::Tile<64, 64, 2, 2, 128, 4, 2>, 0, true, true, false>(GemmDev)"),
            ("vima::gemm_kernel<Tile<128, 128>> act 2", V + "gemm_kernel<unsigned short, vima::(anonymous namespace)::Tile<128, 128, 2, 2, 128, 2, 2>, 2, true, true, false>(GemmDev)"),
            ("vima::gemm_kernel<Tile<128, 128>> act -1", V + "gemm_kernel<float, vima::(anonymous namespace)::Tile<128, 128, 2, 2, 128, 2, 2>, -1, false, true, false>(GemmDev)"),
            ("vima::gemm_resident_kernel<RTile<64, 64, 2, 2, 2>> act 0", V + "gemm_resident_kernel<vima::(anonymous namespace)::RTile<64, 64, 2, 2, 2, false>, 0>(GemmDev)"),
            ("vima::gemm_resident_kernel<RTile<64, 64, 2, 2, 1, true>> (GEGLU pair) act 2", V + "gemm_resident_kernel<vima::(anonymous namespace)::RTile<64, 64, 2, 2, 1, true>, 2>(GemmDev)"),
            ("vima::gemm_persistent_kernel<0, 4>", V + "gemm_persistent_kernel<0, 4, false>(GemmDev)"),
            ("vima::gemm_pp_kernel<0, 5, false>", pp1.replace("<0, 1,", "<0, 5,")), ("vima::gemm_pp_kernel<0, 1, true>", pp1.replace("false", "true"))]
    for a, b in same:
        assert ps._same_kernel(a, b), (a, b)
    differ = [("vima::gemm_pp_kernel<0, 1, false>", pp4), ("vima::gemm_pp_kernel<0, 1, true>", pp1), ("vima::gemm_kernel<Tile<64, 64>> act 0", same[1][1]),
              ("vima::gemm_kernel<Tile<128, 128>> act 0", same[1][1]), ("vima::gemm_resident_kernel<RTile<64, 64, 2, 2, 2>> act 0", same[0][1])]
    for a, b in differ:
        assert not ps._same_kernel(a, b), (a, b)
    # a log that does not match the trace (wrong kernel at position 1) or does not divide it: no attribution, and the note says why
    bad = [log[0], dict(log[0]), log[2]]
    json.dump(bad, open(lp, "w"))
    s2 = ps.summarise(fetch, write, lp)
    assert s2["per_shape"] is None and "mismatch" in s2["per_shape_note"]
    json.dump(log + log[:1], open(lp, "w"))
    s3 = ps.summarise(fetch, write, lp)
    assert s3["per_shape"] is None and "not a multiple" in s3["per_shape_note"]
    # the markdown writer accepts both outcomes
    ps.write_md(s, str(tmp_path / "a.md"))
    ps.write_md(s3, str(tmp_path / "b.md"))
    assert "Per GEMM shape" in open(tmp_path / "a.md").read() and "Per-shape attribution" in open(tmp_path / "b.md").read()
