"""Operator-level parity on a real MI355X: each hand-written HIP kernel against a plain PyTorch fp32 restatement of
the same op (CPU). fp32 mode must agree to fp32 round-off; bf16 mode is compared with a reference that rounds the
OPERANDS to bf16 the same way (products/accumulation fp32 on both sides)."""
import math

import pytest
import torch

from vima_amd import _lib
from tests.gpu_common import bare_policy, bf, max_rel, ptr

pytestmark = pytest.mark.gpu

ACTS = {0: lambda x: x, 1: torch.relu, 2: torch.nn.functional.gelu, 3: lambda x: x * torch.sigmoid(1.702 * x)}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (100, 50, 512), (8, 768, 768), (333, 700, 256), (1030, 2304, 768),
                                   (130, 132, 1536), (1, 100, 512)])
def test_linear_epilogues(prec, variant, M, N, K):
    pol = bare_policy(prec)
    pol.set_option("gemm_variant", variant)
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    for act, use_b, use_m, use_r in [(0, 0, 0, 0), (1, 1, 0, 0), (2, 1, 1, 0), (3, 1, 0, 1), (2, 0, 1, 1)]:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g) if use_b else None
        m = torch.randn(M, N, generator=g) if use_m else None
        r = torch.randn(M, N, generator=g) if use_r else None
        ref = (bf(A) @ bf(W).T) if prec == "bf16" else (A @ W.T)
        if b is not None:
            ref = ref + b
        ref = ACTS[act](ref)
        if m is not None:
            ref = ref * (bf(m) if prec == "bf16" else m)
        if r is not None:
            ref = ref + r
        d = [None if t is None else t.cuda() for t in (A, W, b, m, r)]
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                           ptr(out), pol._stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert max_rel(out, ref) < (2e-3 if prec == "bf16" else 2e-5), (act, use_b, use_m, use_r)


@pytest.mark.parametrize("tile", [0, 1, 2, 7, 8])
@pytest.mark.parametrize("M,N,K", [(520, 768, 768), (1000, 1304, 1536), (77, 520, 128), (8, 768, 768), (30, 200, 3072)])
def test_linear_bf16_output_path(M, N, K, tile):
    """GEMM epilogues that write ONLY the bf16 operand copy (QKV, MLP hidden, K/V projections): 16-byte-store path of the
    LDS-transposed epilogue, with bias / activation / gate multiply / residual."""
    pol = bare_policy("bf16")
    pol.set_option("gemm_tile", tile)
    pol.set_option("op_bf16_out", 1)
    try:
        g = torch.Generator().manual_seed(M + N + K + tile)
        for act, use_b, use_m, use_r in [(0, 0, 0, 0), (3, 1, 0, 0), (2, 1, 1, 0), (0, 1, 0, 1)]:
            A = torch.randn(M, K, generator=g)
            W = torch.randn(N, K, generator=g) * K ** -0.5
            b = torch.randn(N, generator=g) if use_b else None
            m = torch.randn(M, N, generator=g) if use_m else None
            r = torch.randn(M, N, generator=g) if use_r else None
            ref = bf(A) @ bf(W).T
            if b is not None:
                ref = ref + b
            ref = ACTS[act](ref)
            if m is not None:
                ref = ref * bf(m)
            if r is not None:
                ref = ref + r
            ref = bf(ref)
            d = [None if t is None else t.cuda() for t in (A, W, b, m, r)]
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            assert max_rel(out, ref) < 6e-3, (act, use_b, use_m, use_r)   # one extra bf16 rounding of the output
    finally:
        pol.set_option("gemm_tile", 0)
        pol.set_option("op_bf16_out", 0)


@pytest.mark.parametrize("tile,persist", [(0, 1), (1, 1), (2, 0), (2, 1), (8, 1)])
@pytest.mark.parametrize("M,N,K", [(520, 768, 768), (512, 768, 768), (2816, 512, 3072), (64, 768, 768), (8, 768, 3072)])
def test_linear_residual_stream_in_operand_type(M, N, K, tile, persist):
    """The residual-stream form of the T5 / ViT residual GEMMs (round 2): out = bf16(A W^T [+ bias] + bf16 residual), residual
    and output in the operand type (persistent EPI 4 for full 256-tiles, the 8-column LDS epilogue of the one-tile kernels
    otherwise), against torch with the same rounding points."""
    pol = bare_policy("bf16")
    pol.set_option("gemm_tile", tile)
    pol.set_option("gemm_persist", persist)
    pol.set_option("op_stream_T", 1)
    try:
        g = torch.Generator().manual_seed(M + N + K + tile)
        for use_b in (0, 1):
            A = torch.randn(M, K, generator=g)
            W = torch.randn(N, K, generator=g) * K ** -0.5
            b = torch.randn(N, generator=g) if use_b else None
            r = torch.randn(M, N, generator=g) * 3.0
            ref = bf(A) @ bf(W).T
            if b is not None:
                ref = ref + b
            ref = bf(ref + bf(r))
            d = [None if t is None else t.cuda() for t in (A, W, b, None, r)]
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, 0,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            # identical up to one bf16 ulp where the fp32 sums differ in the last bits before rounding
            assert max_rel(out, ref) < 6e-3, (use_b, max_rel(out, ref))
            assert (out.cpu() != ref).float().mean().item() < 0.02
    finally:
        pol.set_option("gemm_tile", 0)
        pol.set_option("gemm_persist", 1)
        pol.set_option("op_stream_T", 0)


@pytest.mark.parametrize("tile,persist", [(2, 0), (2, 1)])
@pytest.mark.parametrize("M,N,K", [(520, 768, 768), (256, 256, 64), (1000, 1300, 3072), (77, 520, 1536), (300, 200, 128),
                                   (512, 768, 768), (2816, 512, 3072), (256, 256, 128)])   # full tiles: persistent kernel
def test_linear_large_tile(M, N, K, tile, persist):
    """The 256x256 / 8-wave tile (bf16) as one-tile-per-workgroup kernel and as the persistent kernel, forced on shapes
    with ragged edges and short K (fewer slices than ring stages falls back to the non-persistent kernel); same
    epilogues."""
    pol = bare_policy("bf16")
    pol.set_option("gemm_tile", tile)
    pol.set_option("gemm_persist", persist)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        for act, use_b, use_m, use_r in [(0, 0, 0, 0), (3, 1, 0, 1), (2, 1, 1, 0), (1, 1, 0, 0)]:
            A = torch.randn(M, K, generator=g)
            W = torch.randn(N, K, generator=g) * K ** -0.5
            b = torch.randn(N, generator=g) if use_b else None
            m = torch.randn(M, N, generator=g) if use_m else None
            r = torch.randn(M, N, generator=g) if use_r else None
            ref = bf(A) @ bf(W).T
            if b is not None:
                ref = ref + b
            ref = ACTS[act](ref)
            if m is not None:
                ref = ref * bf(m)
            if r is not None:
                ref = ref + r
            d = [None if t is None else t.cuda() for t in (A, W, b, m, r)]
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            assert max_rel(out, ref) < 2e-3, (act, use_b, use_m, use_r)
    finally:
        pol.set_option("gemm_tile", 0)
        pol.set_option("gemm_persist", 1)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("mode,B,H,Lq,Lk,D", [(1, 2, 16, 9, 40, 16), (2, 2, 2, 17, 17, 128), (1, 1, 2, 8, 70, 128), (2, 3, 16, 30, 30, 16)])
def test_attention_other_head_dims(prec, mode, B, H, Lq, Lk, D):
    """Head dims 16 and 128 (the head counts are free parameters) run on the exact generic kernel."""
    pol = bare_policy(prec)
    g = torch.Generator().manual_seed(D + Lq)
    q, k, v = (torch.randn(B, L, H, D, generator=g) for L in (Lq, Lk, Lk))
    kmask = torch.rand(B, Lk, generator=g) > 0.2
    kmask[:, 0] = True
    scale = 1.0 / math.sqrt(D)
    ref = attn_ref(bf(q), bf(k), bf(v), kmask, None, scale, mode) if prec == "bf16" else attn_ref(q, k, v, kmask, None, scale, mode)
    out = torch.full((B, Lq, H, D), float("nan"), device="cuda")
    qd, kd, vd, md = q.cuda(), k.cuda(), v.cuda(), kmask.cuda()
    _lib.check(pol._lib.vima_op_attention(pol._handle, ptr(qd), ptr(kd), ptr(vd), ptr(md), None, B, H, Lq, Lk, D, scale, mode, 0,
                                          ptr(out), pol._stream()))
    torch.cuda.synchronize()
    assert max_rel(out, ref) < (1.5e-2 if prec == "bf16" else 1e-5)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("M,N,K", [(8, 768, 3072), (77, 520, 1536), (512, 768, 3072), (300, 1300, 2048)])
def test_linear_split_k(prec, M, N, K):
    """Opt-in deterministic two-pass split-K (underfilled grids, K >= 1536): every epilogue feature goes through the
    reduce pass; two runs are bit-identical."""
    pol = bare_policy(prec)
    pol.set_option("gemm_splitk", 1)
    cast = bf if prec == "bf16" else (lambda t: t)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        for act, use_b, use_m, use_r in [(0, 0, 0, 0), (3, 1, 0, 1), (2, 1, 1, 0), (1, 1, 0, 0)]:
            A = torch.randn(M, K, generator=g)
            W = torch.randn(N, K, generator=g) * K ** -0.5
            b = torch.randn(N, generator=g) if use_b else None
            m = torch.randn(M, N, generator=g) if use_m else None
            r = torch.randn(M, N, generator=g) if use_r else None
            ref = cast(A).double() @ cast(W).double().T
            if b is not None:
                ref = ref + b
            ref = ACTS[act](ref.float())
            if m is not None:
                ref = ref * cast(m)
            if r is not None:
                ref = ref + r
            d = [None if t is None else t.cuda() for t in (A, W, b, m, r)]
            outs = []
            for _ in range(2):
                out = torch.full((M, N), float("nan"), device="cuda")
                _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                                   ptr(out), pol._stream()))
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1])
            assert max_rel(outs[0], ref) < (2e-3 if prec == "bf16" else 2e-5), (act, use_b, use_m, use_r)
    finally:
        pol.set_option("gemm_splitk", 0)


def test_small_tiles_match_128_tile_bitwise():
    """The 64x64 / 32x64 tiles chosen for underfilled grids accumulate K in the same order as the 128x128 tile: results
    must be BIT-identical (batch-composition invariance does not depend on the tile choice)."""
    pol = bare_policy("bf16")
    pol.set_option("gemm_skinny", 0)                  # (M <= 32 otherwise takes gemm_skinny_kernel, whose K split is a different summation order)
    g = torch.Generator().manual_seed(9)
    try:
        for M, N, K in [(8, 768, 3072), (32, 2304, 768), (200, 768, 768), (500, 3072, 768)]:
            A, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
            b, r = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
            outs = []
            for small in (1, 0):
                pol.set_option("gemm_small", small)
                out = torch.full((M, N), float("nan"), device="cuda")
                _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(A), ptr(W), ptr(b), None, ptr(r), M, N, K, 2, ptr(out), pol._stream()))
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (M, N, K)
    finally:
        pol.set_option("gemm_small", 1)
        pol.set_option("gemm_skinny", 1)


def test_pp_kernel_flat_enumeration_is_bit_identical():
    """gemm_pp_kernel drops its XCD raster (A panels padded to a multiple of 8, panel tm on XCD tm % 8) for small grids where the padding costs a
    whole round -- 9 panels x 24 n-tiles, the GEGLU pair of an incremental env step at batch 256 (M = 2304): XCD 0 would get 48 tiles for 32
    workgroups. Same tiles, other workgroups: the output must be BIT-identical with the option off, and right against fp64."""
    pol = bare_policy("bf16")
    pol.set_option("op_bf16_out", 1)
    g = torch.Generator().manual_seed(21)
    try:
        for M, N, K in [(2304, 6144, 768), (2304, 4608, 768), (256 * 17, 3072, 768), (2304, 6144, 3072)]:
            A, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
            b = torch.randn(N, generator=g).cuda()
            outs = []
            for flat in (1, 0):
                pol.set_option("gemm_flat", flat)
                out = torch.full((M, N), float("nan"), device="cuda")
                pol.prof_enable(True)
                _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(A), ptr(W), ptr(b), None, None, M, N, K, 3, ptr(out), pol._stream()))   # QuickGELU
                torch.cuda.synchronize()
                kernels = pol.prof_read_gemm_kernels()
                pol.prof_enable(False)
                assert any(k.startswith("vima::gemm_pp_kernel") for k in kernels), list(kernels)
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (M, N, K)
            ab, wb = A.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
            pre = ab @ wb.T + b.double()
            ref = pre * torch.sigmoid(1.702 * pre)
            assert max_rel(outs[0].double(), ref) < 6e-3, (M, N, K)
    finally:
        pol.set_option("gemm_flat", 1)
        pol.set_option("op_bf16_out", 0)


def test_resident_kernel_matches_ring_tiles_bitwise():
    """gemm_resident_kernel (underfilled grids: (almost) the whole K extent in flight, one barrier per chunk of K-slices) walks K
    in the same order with the same matrix instruction as the ring tiles: every tile shape of it (gemm_tile 10 / 11 / 12 =
    32x32 / 64x32 / 64x64, forced whatever the grid) and the launcher's own choice must be BIT-identical to the ring kernels
    (gemm_resident = 0) -- full, ragged and single-slice shapes, K / 64 not a multiple of the chunk, every epilogue the decoder
    uses (bias, GELU, gate, fp32 residual; bf16-only output; bf16 residual stream)."""
    pol = bare_policy("bf16")
    pol.set_option("gemm_skinny", 0)                  # the launcher's own choice for M <= 32 is gemm_skinny_kernel (test_skinny_kernel_*): not bit-identical by design
    g = torch.Generator().manual_seed(21)
    shapes = [(8, 768, 3072), (9, 768, 768), (32, 2304, 768), (18, 3072, 768), (40, 768, 768), (200, 768, 768), (288, 768, 3072),
              (500, 1536, 768), (33, 100, 192), (5, 36, 64), (64, 96, 320), (300, 200, 448)]
    combos = [(0, False, False, False), (0, True, False, True), (2, True, True, False), (2, False, True, True), (1, True, False, False),
              (3, True, False, True)]
    try:
        for M, N, K in shapes:
            A, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
            b, mu, r = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
            for mode in ("f32out", "bf16out", "streamT"):
                pol.set_option("op_bf16_out", 1 if mode == "bf16out" else 0)
                pol.set_option("op_stream_T", 1 if mode == "streamT" else 0)
                for act, ub, um, ur in combos:
                    if mode == "streamT" and (act != 0 or not ur):
                        continue                      # the bf16-stream epilogue is a residual epilogue without activation
                    outs = {}
                    for name, res, tile, nch in (("ring", 0, 0, 0), ("auto", 1, 0, 0), ("t32x32", 1, 10, 0), ("t64x32", 1, 11, 0), ("t64x64", 1, 12, 0),
                                                 ("t32x32/2buf", 1, 10, 2), ("t64x64/3buf", 1, 12, 3), ("t64x64/5buf", 1, 12, 5)):
                        pol.set_option("gemm_resident", res)
                        pol.set_option("gemm_tile", tile)
                        pol.set_option("gemm_res_nch", nch)
                        out = torch.full((M, N), float("nan"), device="cuda")
                        _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(A), ptr(W), ptr(b) if ub else None, ptr(mu) if um else None,
                                                           ptr(r) if ur else None, M, N, K, act, ptr(out), pol._stream()))
                        torch.cuda.synchronize()
                        outs[name] = out
                    assert not torch.isnan(outs["ring"]).any()
                    for name in [k for k in outs if k != "ring"]:
                        d = (outs[name] - outs["ring"]).abs().max().item()
                        assert torch.equal(outs[name], outs["ring"]), (M, N, K, mode, act, ub, um, ur, name, d)
    finally:
        for k, v in (("gemm_resident", 1), ("gemm_tile", 0), ("gemm_res_nch", 0), ("op_bf16_out", 0), ("op_stream_T", 0), ("gemm_skinny", 1)):
            pol.set_option(k, v)


def test_skinny_kernel_against_fp64_and_the_resident_tile():
    """Round 5: gemm_skinny_kernel (M <= 32: K split over the 4 / 8 / 16 waves of a workgroup, operands straight from global memory in the MFMA
    layout, partial accumulators reduced through LDS) -- every epilogue the decoder uses, all three output modes, K from one k-step per wave to
    192 (three batch sizes of the loop), ragged N, M = 1 .. 32. Not bit-identical to the other tiles (different summation order): compared with
    the fp64 product of the bf16-rounded operands (the tolerance of test_linear_epilogues) and with the resident 32x32 tile to fp32 rounding
    (bf16 outputs: at most one bf16 ulp apart); two launches of the same problem agree bit for bit."""
    pol = bare_policy("bf16")
    g = torch.Generator().manual_seed(77)
    shapes = [(9, 768, 768), (8, 768, 3072), (32, 2304, 768), (1, 512, 768), (18, 3072, 768), (5, 36, 64), (32, 100, 192), (17, 256, 1024), (9, 1024, 2048),
              (3, 64, 4096), (12, 700, 512)]
    combos = [(0, False, False, False), (0, True, False, True), (2, True, True, False), (2, False, True, True), (1, True, False, False), (3, True, False, True)]
    worst = 0.0
    try:
        for M, N, K in shapes:
            A, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
            b, mu, r = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
            for mode in ("f32out", "bf16out", "streamT"):
                pol.set_option("op_bf16_out", 1 if mode == "bf16out" else 0)
                pol.set_option("op_stream_T", 1 if mode == "streamT" else 0)
                for act, ub, um, ur in combos:
                    if mode == "streamT" and (act != 0 or not ur):
                        continue
                    outs = {}
                    for name, sk in (("resident", 0), ("skinny", 1), ("skinny2", 1)):
                        pol.set_option("gemm_skinny", sk)
                        out = torch.full((M, N), float("nan"), device="cuda")
                        pol.prof_enable(True)
                        _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(A), ptr(W), ptr(b) if ub else None, ptr(mu) if um else None,
                                                           ptr(r) if ur else None, M, N, K, act, ptr(out), pol._stream()))
                        torch.cuda.synchronize()
                        kinds = list(pol.prof_read_gemm_kernels())
                        pol.prof_enable(False)
                        assert any("skinny" in k for k in kinds) == bool(sk), (name, kinds)
                        outs[name] = out
                    assert torch.equal(outs["skinny"], outs["skinny2"])
                    ref = bf(A).double() @ bf(W).double().T
                    if ub:
                        ref = ref + b.double()
                    ref = ACTS[act](ref.float())
                    if um:
                        ref = ref * bf(mu)
                    if ur:
                        ref = ref + (bf(r) if mode == "streamT" else r)
                    tol = 2e-3 if mode == "f32out" else 1e-2
                    assert max_rel(outs["skinny"], ref.cuda()) < tol, (M, N, K, mode, act, ub, um, ur, max_rel(outs["skinny"], ref.cuda()))
                    d = max_rel(outs["skinny"], outs["resident"])
                    worst = max(worst, d if mode == "f32out" else 0.0)
                    assert d < (2e-5 if mode == "f32out" else 8e-3), (M, N, K, mode, act, ub, um, ur, d)
        print(f"[skinny] worst fp32-output deviation from the resident tile over {len(shapes)} shapes: {worst:.2e} (relative to max |out|)")
    finally:
        for k, v in (("op_bf16_out", 0), ("op_stream_T", 0), ("gemm_skinny", 1)):
            pol.set_option(k, v)


def test_linear_transpose_detecting():
    """A = I with an asymmetric W: output must equal W^T exactly (catches swapped C-layout / operand order)."""
    pol = bare_policy("fp32")
    n = 128
    A = torch.eye(n)
    W = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    out = torch.empty(n, n, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()
    _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(Ad), ptr(Wd), None, None, None, n, n, n, 0, ptr(out), pol._stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), W.T)


@pytest.mark.parametrize("rows,E,rms", [(7, 256, 0), (1000, 768, 0), (513, 384, 1), (64, 1024, 0), (5, 320, 1)])
def test_layernorm(rows, E, rms):
    pol = bare_policy("fp32")
    g = torch.Generator().manual_seed(rows + E)
    x = torch.randn(rows, E, generator=g) * 3 + 0.5
    ga, be = torch.randn(E, generator=g), torch.randn(E, generator=g)
    if rms:
        ref = ga * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    else:
        ref = torch.nn.functional.layer_norm(x, (E,), ga, be, 1e-5)
    out = torch.empty(rows, E, device="cuda")
    xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()
    _lib.check(pol._lib.vima_op_layernorm(pol._handle, ptr(xd), ptr(gd), None if rms else ptr(bd), 1e-6 if rms else 1e-5, rms,
                                          rows, E, ptr(out), pol._stream()))
    torch.cuda.synchronize()
    assert max_rel(out, ref) < 1e-5


@pytest.mark.parametrize("rows,E,rms", [(1, 768, 0), (2, 768, 0), (7, 256, 0), (1001, 768, 0), (40961, 768, 0), (513, 512, 1), (64, 1024, 0),
                                        (5, 320, 1), (33, 384, 0)])
def test_layernorm_of_a_bf16_stream(rows, E, rms):
    """The ViT / decoder LayerNorms read the residual stream in bf16 (stream_T): rows of 256 n elements take the half-wave-per-row
    kernel (two rows per wave, odd row counts, a grid-strided walk from 16 384 rows on), other lengths the one-wave-per-row kernel;
    both against torch on the bf16-rounded input."""
    pol = bare_policy("bf16")
    g = torch.Generator().manual_seed(rows * 7 + E)
    x = bf(torch.randn(rows, E, generator=g) * 3 + 0.5)
    ga, be = torch.randn(E, generator=g), torch.randn(E, generator=g)
    if rms:
        ref = ga * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    else:
        ref = torch.nn.functional.layer_norm(x, (E,), ga, be, 1e-5)
    out = torch.empty(rows, E, device="cuda")
    xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()
    pol.set_option("op_stream_T", 1)
    try:
        _lib.check(pol._lib.vima_op_layernorm(pol._handle, ptr(xd), ptr(gd), None if rms else ptr(bd), 1e-6 if rms else 1e-5, rms,
                                              rows, E, ptr(out), pol._stream()))
        torch.cuda.synchronize()
    finally:
        pol.set_option("op_stream_T", 0)
    assert max_rel(out, ref) < 1e-5
    assert torch.isfinite(out).all()


def attn_ref(q, k, v, kmask, relbias, scale, mode):
    """Literal restatement of the three score pipelines (prompt_encoder.py:769-801, components.py:184-207, :54-69)."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q, k)
    fmin = torch.finfo(torch.float32).min
    madd = (1.0 - kmask[:, None, None, :].float()) * fmin
    if mode == 0:
        idx = (torch.arange(Lk)[None, :] - torch.arange(Lq)[:, None]) + Lk - 1
        s = s + (relbias[:, idx][None] + madd)
    elif mode == 1:
        s = s * scale + madd
    else:
        tri = torch.tril(torch.ones(Lq, Lk))
        s = (s * scale) * tri + -1e4 * (1 - tri)
        s = s + madd
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, dim=-1), v)


ATTN_CASES = [(0, 2, 12, 64, 64, 64), (0, 1, 12, 100, 100, 64), (1, 3, 8, 8, 40, 32), (1, 2, 24, 71, 300, 32),
              (2, 2, 8, 9, 9, 32), (2, 3, 24, 71, 71, 32), (1, 2, 4, 5, 33, 64), (2, 1, 4, 40, 40, 64),
              (0, 2, 12, 300, 300, 64), (0, 1, 3, 512, 512, 64), (2, 2, 4, 200, 200, 32), (1, 2, 4, 130, 65, 64),
              # few queries, many keys: the split-key kernel (Lq <= 32, Lk >= 64)
              (1, 3, 8, 8, 200, 32), (1, 2, 4, 30, 512, 64), (1, 2, 24, 8, 512, 32), (1, 2, 4, 32, 64, 32), (1, 2, 2, 1, 129, 64)]


@pytest.mark.parametrize("prec,impl", [("fp32", 0), ("bf16", 0), ("bf16", 1), ("bf16", 4)])
@pytest.mark.parametrize("mode,B,H,Lq,Lk,D", ATTN_CASES)
def test_attention(prec, impl, mode, B, H, Lq, Lk, D):
    """impl 0 = exact generic kernel, 1 = MFMA flash (1-wave kernel below 64 queries, 4-wave LDS-shared kernel above),
    4 = MFMA flash with the 4-wave kernel forced for every shape."""
    pol = bare_policy(prec)
    pol.set_option("attn4_min_lq", 1 if impl == 4 else 64)
    impl = 1 if impl == 4 else impl
    g = torch.Generator().manual_seed(mode * 100 + Lq + Lk)
    sc = 1.0 if mode else 0.4
    q = torch.randn(B, Lq, H, D, generator=g) * sc
    k = torch.randn(B, Lk, H, D, generator=g) * sc
    v = torch.randn(B, Lk, H, D, generator=g)
    kmask = torch.rand(B, Lk, generator=g) > 0.2
    kmask[:, 0] = True
    if B > 1 and mode == 1:
        kmask[1, :] = False   # an all-masked row degenerates to the uniform distribution in the reference
    relbias = torch.randn(H, 2 * Lk - 1, generator=g) if mode == 0 else None
    scale = 1.0 if mode == 0 else 1.0 / math.sqrt(D)
    ref = attn_ref(bf(q), bf(k), bf(v), kmask, relbias, scale, mode) if prec == "bf16" else attn_ref(q, k, v, kmask, relbias, scale, mode)
    out = torch.full((B, Lq, H, D), float("nan"), device="cuda")
    qd, kd, vd, md = q.cuda(), k.cuda(), v.cuda(), kmask.cuda()
    rd = relbias.cuda() if relbias is not None else None
    _lib.check(pol._lib.vima_op_attention(pol._handle, ptr(qd), ptr(kd), ptr(vd), ptr(md), ptr(rd), B, H, Lq, Lk, D, scale, mode,
                                          impl, ptr(out), pol._stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    # bf16: probabilities and outputs are rounded to bf16 (2^-9 relative each)
    assert max_rel(out, ref) < (1.5e-2 if prec == "bf16" else 1e-5)


@pytest.mark.parametrize("B,H,L,D,qg", [(2, 12, 512, 64, 1), (1, 3, 1030, 64, 2), (2, 4, 300, 32, 1), (1, 12, 200, 64, 1)])
def test_t5_attention_constant_bias_beyond_the_last_bucket(B, H, L, D, qg):
    """Round 4: the bucketed T5 bias table is constant from |key - query| = 91 on; with that promise (AttnArgs::bias_far, here the test
    option op_bias_far) key tiles wholly beyond it take the constant as a scalar instead of per-score LDS reads. Same values, same arithmetic:
    the outputs must be IDENTICAL to the run without the promise (and agree with torch); ragged sizes, masked keys, both wave geometries."""
    pol = bare_policy("bf16")
    lib = pol._lib
    g = torch.Generator().manual_seed(L + D)
    q = torch.randn(B, L, H, D, generator=g) * 0.4
    k = torch.randn(B, L, H, D, generator=g) * 0.4
    v = torch.randn(B, L, H, D, generator=g)
    kmask = torch.rand(B, L, generator=g) > 0.1
    kmask[:, 0] = True
    emb = torch.randn(32, H, generator=g)                                  # relative_attention_bias.weight [buckets, heads]
    delta = torch.arange(-(L - 1), L)
    bucket = torch.tensor([lib.vima_t5_bucket(int(d)) for d in delta.tolist()])
    relbias = emb[bucket].T.contiguous()                                   # [H, 2L - 1], index (j - i) + L - 1
    far = next(n for n in range(1, L) if bool((bucket[L - 1 + n:] == bucket[-1]).all()) and bool((bucket[:L - n] == bucket[0]).all()))
    assert far == 91 or L <= 91
    ref = attn_ref(bf(q), bf(k), bf(v), kmask, relbias, 1.0, 0)
    qd, kd, vd, md, rd = q.cuda(), k.cuda(), v.cuda(), kmask.cuda(), relbias.cuda()
    pol.set_option("attn_qg", qg)
    outs = []
    try:
        for promise in (0, far):
            pol.set_option("op_bias_far", promise)
            out = torch.full((B, L, H, D), float("nan"), device="cuda")
            _lib.check(lib.vima_op_attention(pol._handle, ptr(qd), ptr(kd), ptr(vd), ptr(md), ptr(rd), B, H, L, L, D, 1.0, 0, 1, ptr(out), pol._stream()))
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        pol.set_option("op_bias_far", 0)
        pol.set_option("attn_qg", 1)
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1])
    assert max_rel(outs[1], ref) < 1.5e-2


@pytest.mark.parametrize("mode,B,H,Lq,Lk,D", [(0, 1, 12, 300, 300, 64), (0, 2, 3, 1030, 1030, 64), (1, 2, 24, 300, 520, 32), (1, 1, 4, 256, 64, 64),
                                             (0, 1, 2, 257, 257, 32)])
@pytest.mark.parametrize("qg", [1, 2])
def test_attention_lds_dma_ring_both_geometries(qg, mode, B, H, Lq, Lk, D):
    """The 4-wave flash kernel stages K / V by LDS-DMA into a ring of 2 stages (32 queries per wave, default) or 3 stages (64 queries
    per wave, option attn_qg = 2, from 256 queries on): ragged last tiles, a single tile, more tiles than stages, masked keys; the two
    geometries process every query row with the same arithmetic, so their outputs are IDENTICAL, and both agree with torch."""
    pol = bare_policy("bf16")
    g = torch.Generator().manual_seed(mode * 100 + Lq + Lk)
    sc = 1.0 if mode else 0.4
    q = torch.randn(B, Lq, H, D, generator=g) * sc
    k = torch.randn(B, Lk, H, D, generator=g) * sc
    v = torch.randn(B, Lk, H, D, generator=g)
    kmask = torch.rand(B, Lk, generator=g) > 0.2
    kmask[:, 0] = True
    relbias = torch.randn(H, 2 * Lk - 1, generator=g) if mode == 0 else None
    scale = 1.0 if mode == 0 else 1.0 / math.sqrt(D)
    ref = attn_ref(bf(q), bf(k), bf(v), kmask, relbias, scale, mode)
    qd, kd, vd, md = q.cuda(), k.cuda(), v.cuda(), kmask.cuda()
    rd = relbias.cuda() if relbias is not None else None
    outs = {}
    try:
        for geo in (qg, 3 - qg):
            pol.set_option("attn_qg", geo)
            out = torch.full((B, Lq, H, D), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_attention(pol._handle, ptr(qd), ptr(kd), ptr(vd), ptr(md), ptr(rd), B, H, Lq, Lk, D, scale, mode,
                                                  1, ptr(out), pol._stream()))
            torch.cuda.synchronize()
            outs[geo] = out.cpu()
    finally:
        pol.set_option("attn_qg", 1)
    assert torch.isfinite(outs[qg]).all()
    assert max_rel(outs[qg], ref) < 1.5e-2
    assert torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("M,N,K,act,mode", [(16384, 768, 768, 0, "stream"), (16384, 768, 768, 0, "res32"), (16384, 2304, 128, 0, "bf16"),
                                             (16384, 3072, 768, 3, "bf16"), (24576, 1536, 3072, 1, "bf16"), (32768, 768, 3072, 0, "stream"),
                                             (12288, 1024, 256, 2, "gate"), (65536, 256, 1152, 0, "bf16")])
def test_linear_pingpong_is_bit_identical(M, N, K, act, mode):
    """The ping-pong (8-phase) main loop of the persistent 256x256 GEMM (option gemm_pp, default on) accumulates K in the
    same order as the round-2 loop and shares its epilogue: identical bits for every specialised epilogue (bf16-only
    output, GEGLU gate, fp32 residual, bf16 residual stream), several tiles per workgroup, short and long K, and the usual
    agreement with torch."""
    pol = bare_policy("bf16")
    pol.set_option("op_bf16_out", 0 if mode == "res32" else 1)
    pol.set_option("op_stream_T", 1 if mode == "stream" else 0)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g) * 3.0 if mode in ("stream", "res32") else None
        mul = torch.randn(M, N, generator=g) if mode == "gate" else None
        d = [None if t is None else t.cuda() for t in (A, W, b, mul, r)]
        outs = []
        for pp in (0, 1):
            pol.set_option("gemm_pp", pp)
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.isfinite(outs[1]).all()
        assert torch.equal(outs[0], outs[1])
        ref = bf(A) @ bf(W).T + b
        if act == 1:
            ref = torch.relu(ref)
        elif act == 2:
            ref = torch.nn.functional.gelu(ref)
        elif act == 3:
            ref = ref * torch.sigmoid(1.702 * ref)
        if mode == "gate":
            ref = bf(ref * bf(mul))
        elif mode == "stream":
            ref = bf(ref + bf(r))
        elif mode == "res32":
            ref = ref + r
        else:
            ref = bf(ref)
        assert max_rel(outs[1], ref) < 6e-3, max_rel(outs[1], ref)
    finally:
        pol.set_option("gemm_pp", 1)
        pol.set_option("op_bf16_out", 0)
        pol.set_option("op_stream_T", 0)

@pytest.mark.parametrize("M,N,K,act,mode", [(131072, 768, 768, 0, "stream"), (65536, 768, 3072, 0, "stream"), (65536, 768, 768, 0, "res32"),
                                             (32768, 3072, 768, 2, "gate"), (65536, 768, 768, 0, "out32")])
def test_persistent_epilogue_under_full_load_matches_the_one_tile_kernel_bitwise(M, N, K, act, mode):
    """Round 4: gfx950 does not retire loads and stores in issue order against each other, so the persistent kernels' epilogues with a per-row
    operand (bf16 / fp32 residual, GEGLU gate) wait for younger LOADS only and issue a slab's stores one slab late. The failure mode of a wrong
    count is a slab finished with an operand that has not landed yet, and it only shows when every CU runs its epilogue at once -- so: the
    benchmark's own shapes at (nearly) full size, several tiles per CU, repeated, bit for bit against the one-tile-per-workgroup kernel, whose
    epilogue takes its operands with compiler-tracked loads."""
    pol = bare_policy("bf16")
    pol.set_option("op_bf16_out", 0 if mode in ("res32", "out32") else 1)     # out32: fp32 output WITHOUT a residual (the ViT's patch-embed GEMM)
    pol.set_option("op_stream_T", 1 if mode == "stream" else 0)
    try:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g, device="cuda")
        W = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
        b = torch.randn(N, generator=g, device="cuda")
        r = torch.randn(M, N, generator=g, device="cuda") * 3.0 if mode in ("stream", "res32") else None
        mul = torch.randn(M, N, generator=g, device="cuda") if mode == "gate" else None
        out = torch.empty(M, N, device="cuda")

        def run():
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(A), ptr(W), ptr(b), ptr(mul), ptr(r), M, N, K, act, ptr(out), pol._stream()))
            torch.cuda.synchronize()
            return out.clone()

        pol.set_option("gemm_persist", 0)          # one 256x256 tile per workgroup (gemm_kernel)
        ref = run()
        pol.set_option("gemm_persist", 1)
        for pp in (1, 0):                          # ping-pong and round-2 persistent kernels share the epilogue
            pol.set_option("gemm_pp", pp)
            for _ in range(3):
                got = run()
                bad = int((got != ref).sum())
                assert bad == 0, f"gemm_pp={pp}: {bad} of {got.numel()} elements differ from the one-tile kernel"
        assert bool(torch.isfinite(ref).all())
    finally:
        pol.set_option("gemm_pp", 1)
        pol.set_option("gemm_persist", 1)
        pol.set_option("op_bf16_out", 0)
        pol.set_option("op_stream_T", 0)


@pytest.mark.parametrize("M,N,K,act,stream", [(16384, 768, 768, 0, 1), (16384, 768, 768, 0, 0), (16384, 2304, 128, 0, 0), (16384, 3072, 768, 3, 0),
                                              (24576, 1536, 3072, 1, 0), (32768, 768, 3072, 0, 1)])
def test_linear_wide_tile_is_bit_identical(M, N, K, act, stream):
    """The 256x384 persistent tile (option gemm_wide; N a multiple of 384, bf16-only output or bf16 residual stream) accumulates
    K in the same order as every other tile shape: identical bits to the 256x256 kernels, and the usual agreement with torch."""
    pol = bare_policy("bf16")
    pol.set_option("op_bf16_out", 1)
    pol.set_option("op_stream_T", stream)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g) * 3.0 if stream else None
        d = [None if t is None else t.cuda() for t in (A, W, b, None, r)]
        outs = []
        for wide in (0, 1):
            pol.set_option("gemm_wide", wide)
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.isfinite(outs[1]).all()
        assert torch.equal(outs[0], outs[1])
        ref = bf(A) @ bf(W).T + b
        if act == 1:
            ref = torch.relu(ref)
        elif act == 3:
            ref = ref * torch.sigmoid(1.702 * ref)
        ref = bf(ref + bf(r)) if stream else bf(ref)
        assert max_rel(outs[1], ref) < 6e-3, max_rel(outs[1], ref)
    finally:
        pol.set_option("gemm_wide", 0)
        pol.set_option("op_bf16_out", 0)
        pol.set_option("op_stream_T", 0)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("M,N,K,act,stream", [(16384, 768, 768, 0, 1), (16384, 768, 768, 0, 0), (16384, 2304, 128, 0, 0), (16384, 3072, 768, 3, 0),
                                              (24576, 1536, 3072, 1, 0), (32768, 768, 3072, 0, 1), (8192, 6144, 256, 0, 0)])
def test_linear_q4_tile_is_bit_identical(M, N, K, act, stream, mode):
    """gemm_q4_kernel (option gemm_q4: 256x384 tile on four waves, one per SIMD; Gray-code quadrant phases, rolling in-place fragment reloads,
    inline-asm MFMAs with 128 of the 384 accumulator registers in VGPRs) accumulates K in the same order with the same MFMA as every other tile
    shape: identical bits to the 256x256 ping-pong kernel (bias, activation, bf16 residual stream with its RMS partials), and the usual
    agreement with torch. The shapes cover one to sixteen tiles per workgroup and 2 .. 48 K-tiles. mode 1: the 256x384 tile, mode 2: the 128x384 tile
    (wave tile 64x192, its own LDS layout and wait counts)."""
    pol = bare_policy("bf16")
    pol.set_option("op_bf16_out", 1)
    pol.set_option("op_stream_T", stream)
    try:
        g = torch.Generator().manual_seed(M + N + K + 1)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g) * 3.0 if stream else None
        d = [None if t is None else t.cuda() for t in (A, W, b, None, r)]
        outs, kinds = [], []
        for q4 in (0, mode):
            pol.set_option("gemm_q4", q4)
            out = torch.full((M, N), float("nan"), device="cuda")
            pol.prof_enable(True)
            _lib.check(pol._lib.vima_op_linear(pol._handle, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), M, N, K, act,
                                               ptr(out), pol._stream()))
            torch.cuda.synchronize()
            kinds.append([l["kernel"] for l in pol.prof_read_gemm_launches()])
            pol.prof_enable(False)
            outs.append(out)
        assert any("gemm_q4_kernel" in k and k.endswith(f", {3 - mode}>") for k in kinds[1]) and not any("gemm_q4_kernel" in k for k in kinds[0]), kinds
        assert torch.isfinite(outs[1]).all()
        assert torch.equal(outs[0], outs[1])
        ref = bf(A) @ bf(W).T + b
        if act == 1:
            ref = torch.relu(ref)
        elif act == 3:
            ref = ref * torch.sigmoid(1.702 * ref)
        ref = bf(ref + bf(r)) if stream else bf(ref)
        assert max_rel(outs[1], ref) < 6e-3, max_rel(outs[1], ref)
    finally:
        pol.set_option("gemm_q4", 6)   # the default: the 256x384 tile from 32 768 rows on
        pol.set_option("op_bf16_out", 0)
        pol.set_option("op_stream_T", 0)
