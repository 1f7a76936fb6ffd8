"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (vima_amd/).

Runs the reference's OWN evaluation loop -- `main()` of /root/reference/scripts/example.py:78-240, together with its
`prepare_prompt` (:243-371) and `prepare_obs` (:374-473) -- against any policy object, without VIMA-Bench, gym, OpenCV or
the Hub tokenizer: the three function bodies are compiled from the reference's source text with `ast` (nothing is copied
into this repository; the module itself cannot be imported) and executed in a namespace in which
  * `create_policy_from_ckpt`                      -> the policy under test (wrapped in a recorder),
  * `make` / `TimeLimitWrapper` / `ResetFaultToleranceWrapper` -> a synthetic environment (random frames + segmentation maps
    from `oracle/preprocess_oracle.synthetic_frames`, a fixed prompt with object and scene placeholders, a fixed episode
    length) -- the simulator is out of scope,
  * `tokenizer`                                    -> a whitespace tokenizer with fixed ids (the t5-base vocabulary needs the Hub),
  * `cv2.resize(.., INTER_AREA)`                   -> `oracle/preprocess_oracle.resize_area_32` (OpenCV is not installed),
  * `np.bool` (removed in numpy 1.24, used at example.py:327,347) -> `bool`.
`RecordingPolicy` logs every policy call of the loop with its tensor arguments and results; the same recorder around the
repository's own swap-in loop (`examples/reference_loop.py`) must produce the identical log (tests/test_eval_loop.py),
which is what lets the GPU test run that loop on the box where /root/reference does not exist."""
from __future__ import annotations

import ast
import collections.abc
import os
import sys
import types

import numpy as np
import torch

from . import ref_shim
from .preprocess_oracle import resize_area_32, synthetic_frames

VIEWS = ("front", "top")
PLACEHOLDER_NAMES = ("dragged_obj", "base_obj", "scene")
PROMPT = "put the {dragged_obj} into the {base_obj} as in {scene} please ."


class EpisodeLimit(Exception):
    """Raised by the synthetic environment when the loop asks for a second episode (main() loops `while True`)."""


class FixedTokenizer:
    """Whitespace tokenizer with fixed ids: words get ids from a small table, placeholders keep their text (the reference
    recognises them by `token in PLACEHOLDERS`, example.py:251-255); `encode` appends the t5 end-of-sequence id 1 like
    `add_special_tokens=True` does."""

    WORDS = {"put": 474, "the": 8, "into": 139, "as": 38, "in": 16, "please": 754, ".": 5}

    class _Enc:
        def __init__(self, ids, tokens):
            self.ids, self.tokens = ids, tokens

    def encode(self, prompt, add_special_tokens=True):
        toks = prompt.split()
        ids = [self.WORDS.get(t, 32000 + i) for i, t in enumerate(toks)]
        if add_special_tokens:
            toks, ids = toks + ["</s>"], ids + [1]
        return FixedTokenizer._Enc(ids, toks)


def placeholders():
    return ["{" + n + "}" for n in PLACEHOLDER_NAMES]


class SyntheticEnv:
    """Stands in for `TimeLimitWrapper(ResetFaultToleranceWrapper(make(...)))`: the attributes and methods main() touches
    (example.py:82-112, 236-239). Observations are single frames (no time axis) like VIMA-Bench's; object ids 2 .. n_obj+1."""

    def __init__(self, n_steps=6, n_obj=4, seed=0, H=128, W=256, missing_at=None):
        self.n_steps, self.n_obj, self.seed, self.H, self.W = n_steps, n_obj, seed, H, W
        self.missing_at = missing_at or {}          # step -> object ids that are not visible in that step
        self.global_seed = None
        self.actions = []
        self.resets = 0
        ids = list(range(2, 2 + n_obj))
        self.meta_info = {
            "n_objects": n_obj,
            "obj_id_to_info": {i: {"obj_name": f"obj{i}"} for i in ids},
            "action_bounds": {"low": np.array([0.25, -0.5], dtype=np.float32), "high": np.array([0.75, 0.5], dtype=np.float32)},
        }
        self.prompt = PROMPT
        self.prompt_assets = self._assets(ids)
        self._t = 0

    def _frame(self, t):
        out = {"rgb": {}, "segm": {}}
        for i, v in enumerate(VIEWS):
            rgb, segm, _ = synthetic_frames(1, self.n_obj, H=self.H, W=self.W, seed=self.seed * 1000 + 10 * t + i,
                                            missing=self.missing_at.get(t, ()))
            out["rgb"][v], out["segm"][v] = rgb[0], segm[0]
        out["ee"] = t % 2
        return out

    def _assets(self, ids):
        """Prompt assets: an "object" placeholder must be visible (>= 2 px in both directions) in BOTH views and a "scene"
        must show at least one object per view -- the reference's prepare_prompt cannot represent an asset without objects
        (np.asarray([]) is 1-D, the padding concat at example.py:360 then fails), so frames are redrawn until that holds."""
        def visible(segm, oid):
            ys, xs = np.nonzero(segm == oid)
            return len(xs) >= 2 and len(ys) >= 2

        assets = {}
        for k, name in enumerate(PLACEHOLDER_NAMES):
            oid = ids[k % len(ids)]
            for attempt in range(64):
                rgb, segm = {}, {}
                for i, v in enumerate(VIEWS):
                    r, s, _ = synthetic_frames(1, self.n_obj, H=self.H, W=self.W, seed=self.seed * 1000 + 500 + 10 * k + i + 97 * attempt)
                    rgb[v], segm[v] = r[0], s[0]
                if name == "scene":
                    ok = all(any(visible(segm[v], i) for i in ids) for v in VIEWS)
                else:
                    ok = all(visible(segm[v], oid) for v in VIEWS)
                if ok:
                    break
            else:
                raise RuntimeError("could not draw a visible prompt asset")
            if name == "scene":
                segm["obj_info"] = [{"obj_id": i} for i in ids]
                kind = "scene"
            else:
                segm["obj_info"] = {"obj_id": oid}
                kind = "object"
            assets[name] = {"rgb": rgb, "segm": segm, "placeholder_type": kind}
        return assets

    def reset(self):
        self.resets += 1
        if self.resets > 1:
            raise EpisodeLimit()
        self._t = 0
        return self._frame(0)

    def render(self):
        pass

    def step(self, actions):
        self.actions.append({k: np.asarray(v).copy() for k, v in actions.items()})
        self._t += 1
        return self._frame(self._t), 0.0, self._t >= self.n_steps, {}


class _NumpyCompat:
    """numpy with the alias the reference still uses (`np.bool`, example.py:327,347)."""
    bool = bool

    def __getattr__(self, name):
        return getattr(np, name)


def _to_cpu(x):
    if torch.is_tensor(x):
        return x.detach().to("cpu").clone()
    if isinstance(x, collections.abc.Mapping):            # dict, the reference's DataDict, vima_amd's MapDict
        return {k: _to_cpu(x[k]) for k in x.keys()}
    if isinstance(x, (list, tuple)):
        return [_to_cpu(v) for v in x]
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x.copy())
    return x


class RecordingPolicy:
    """Forwards every attribute to `policy`; the calls the evaluation loop makes are logged as (name, args, result) with all
    tensors copied to the CPU. Action distributions are logged through their modes (the only thing the loop reads)."""

    LOGGED = ("forward_prompt_assembly", "forward_obs_token", "forward", "forward_action_decoder", "forward_action_token",
              "_de_discretize_actions")

    def __init__(self, policy):
        self._policy = policy
        self.log = []

    def __getattr__(self, name):
        target = getattr(self._policy, name)
        if name not in RecordingPolicy.LOGGED:
            return target

        def call(*args, **kwargs):
            out = target(*args, **kwargs)
            rec_out = {k: v.mode() for k, v in out.items()} if name == "forward_action_decoder" else out
            self.log.append((name, _to_cpu(list(args)), _to_cpu(dict(kwargs)), _to_cpu(rec_out)))
            return out
        return call


def _flatten(x, prefix=""):
    if torch.is_tensor(x):
        return {prefix: x}
    if isinstance(x, dict):
        out = {}
        for k in sorted(x.keys(), key=str):
            out.update(_flatten(x[k], f"{prefix}.{k}"))
        return out
    if isinstance(x, (list, tuple)):
        out = {}
        for i, v in enumerate(x):
            out.update(_flatten(v, f"{prefix}[{i}]"))
        return out
    return {prefix: x}


def compare_logs(a, b, atol=0.0):
    """Call-by-call comparison of two RecordingPolicy logs: same methods in the same order, same argument / result structure,
    integer and boolean tensors identical, floating-point tensors within `atol` (0: identical). Returns the number of
    compared tensors; raises AssertionError naming the first difference."""
    assert [c[0] for c in a] == [c[0] for c in b], ([c[0] for c in a], [c[0] for c in b])
    n = 0
    for ci, (ca, cb) in enumerate(zip(a, b)):
        for part, (xa, xb) in zip(("args", "kwargs", "result"), zip(ca[1:], cb[1:])):
            fa, fb = _flatten(xa), _flatten(xb)
            assert list(fa.keys()) == list(fb.keys()), (ci, ca[0], part, list(fa.keys()), list(fb.keys()))
            for k in fa:
                va, vb = fa[k], fb[k]
                where = f"call {ci} {ca[0]} {part}{k}"
                if torch.is_tensor(va):
                    assert torch.is_tensor(vb) and va.shape == vb.shape and va.dtype == vb.dtype, (where, va.shape, getattr(vb, "shape", None), va.dtype, getattr(vb, "dtype", None))
                    if va.dtype.is_floating_point and atol > 0:
                        assert (va - vb).abs().max().item() <= atol if va.numel() else True, (where, (va - vb).abs().max().item())
                    else:
                        assert torch.equal(va, vb), where
                    n += 1
                else:
                    assert va == vb, (where, va, vb)
    return n


def load_reference_loop(policy, env, device="cpu"):
    """-> (main, cfg): the reference's `main` bound to `policy` (returned by its `create_policy_from_ckpt`) and `env`
    (returned by its `make`); call `main(cfg)` and catch EpisodeLimit."""
    path = os.path.join(ref_shim.REFERENCE_ROOT, "scripts", "example.py")
    tree_ = ast.parse(open(path).read())
    want = ("main", "prepare_prompt", "prepare_obs")
    fns = [n for n in tree_.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert sorted(f.name for f in fns) == sorted(want)
    ref_shim.load_reference()
    if "omegaconf" not in sys.modules:                # any_to_datadict imports it lazily (vima/utils.py:650)
        om = types.ModuleType("omegaconf")
        om.OmegaConf = type("OmegaConf", (), {})
        om.DictConfig = type("DictConfig", (), {})
        sys.modules["omegaconf"] = om
    import vima.utils as vu
    from einops import rearrange
    ns = {k: getattr(vu, k) for k in dir(vu) if not k.startswith("_")}          # `from vima.utils import *` (example.py:10)
    ns.update({
        "np": _NumpyCompat(), "torch": torch, "rearrange": rearrange, "os": os,
        "cv2": types.SimpleNamespace(INTER_AREA=3, resize=lambda img, size, interpolation: resize_area_32(np.ascontiguousarray(img))),
        "tokenizer": FixedTokenizer(), "PLACEHOLDERS": placeholders(),
        "ALL_PARTITIONS": ["synthetic"], "PARTITION_TO_SPECS": {"test": {"synthetic": {"synthetic_task": {}}}},
        "create_policy_from_ckpt": lambda ckpt, dev: policy,
        "make": lambda *a, **k: env, "ResetFaultToleranceWrapper": lambda e: e, "TimeLimitWrapper": lambda e, bonus_steps: e,
    })
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    cfg = types.SimpleNamespace(partition="synthetic", task="synthetic_task", ckpt="<synthetic>", device=device)
    return ns["main"], cfg


def run_reference_loop(policy, env, device="cpu"):
    """One episode of the reference's main() -> the recorder's log (the actions the loop sent to the environment are in
    `env.actions`)."""
    rec = RecordingPolicy(policy)
    main, cfg = load_reference_loop(rec, env, device)
    try:
        main(cfg)
    except EpisodeLimit:
        pass
    return rec.log


# ---- CPU twins of vima_amd.preprocess.prepare_obs / prepare_prompt_images (same signatures and output structure), built on
# the numpy oracle: they let `examples/reference_loop.run_episode` run in a container without a GPU. The GPU functions are
# checked bit-exact against the same oracle in tests/test_preprocess_gpu.py.
def cpu_prepare_obs(*, obs, rgb_dict=None, meta, device=None):
    from vima_amd.containers import MapDict
    from .preprocess_oracle import prepare_obs_oracle
    assert not (rgb_dict is not None and "rgb" in obs)
    rgb = rgb_dict or obs.pop("rgb")
    segm = obs.pop("segm")
    assert meta["n_objects"] == len(meta["obj_id_to_info"])
    out = prepare_obs_oracle({v: np.asarray(rgb[v]) for v in rgb}, {v: np.asarray(segm[v]) for v in segm}, list(meta["obj_id_to_info"].keys()))
    objs = MapDict({k: MapDict({v: torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(1) for v, a in d.items()}) for k, d in out.items()})
    return {"ee": torch.as_tensor(np.asarray(obs["ee"])).to(torch.int64).reshape(-1, 1), "objects": objs}


def cpu_prepare_prompt_images(prompt_assets, names, views=VIEWS, device=None):
    from vima_amd.containers import MapDict
    from .preprocess_oracle import crop_objects_view
    views = sorted(views)
    per = {v: [] for v in views}
    for name in names:
        asset = prompt_assets[name]
        info = asset["segm"]["obj_info"]
        ids = [info["obj_id"]] if asset["placeholder_type"] == "object" else [e["obj_id"] for e in info]
        for v in views:
            c, b, m = crop_objects_view(np.asarray(asset["rgb"][v]), np.asarray(asset["segm"][v]), ids)
            n = int(m.sum())                                 # visible objects first (reference order), invisible dropped
            per[v].append((torch.from_numpy(c[:n].copy()), torch.from_numpy(b[:n].copy()), torch.from_numpy(m[:n].copy())))
    out = {"cropped_img": {}, "bbox": {}, "mask": {}}
    for v in views:
        mx = max(c.shape[0] for c, _, _ in per[v])
        out["cropped_img"][v] = torch.stack([torch.cat([c, c.new_zeros(mx - c.shape[0], 3, 32, 32)]) for c, _, _ in per[v]])
        out["bbox"][v] = torch.stack([torch.cat([b, b.new_zeros(mx - b.shape[0], 4)]) for _, b, _ in per[v]])
        out["mask"][v] = torch.stack([torch.cat([m, m.new_zeros(mx - m.shape[0])]) for _, _, m in per[v]])
    return MapDict({k: MapDict(d) for k, d in out.items()})
