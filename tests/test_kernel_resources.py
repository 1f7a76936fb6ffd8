"""Compile-time resource guard (no GPU): the hot kernels must not use scratch memory. hipcc once kept a staging array
of the attention kernels in scratch (a vmcnt(0) + scratch round trip behind every global load) and spilled the
all-runtime persistent GEMM epilogue -- both cost 5-10 % silently. Cross-compiles the two sources for gfx950 with
-Rpass-analysis=kernel-resource-usage and checks ScratchSize / VGPR budget of the kernels the benchmark runs on."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


_compiled = {}


def _compile(src):
    """One device-only compile per source: assembly on stdout, the resource-usage remarks on stderr (both tests of gemm.hip
    share it)."""
    if src not in _compiled:
        out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
                              "--cuda-device-only", "-S", os.path.join(ROOT, "vima_amd", "csrc", src), "-o", "-",
                              "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        _compiled[src] = (out.stdout, out.stderr)
    return _compiled[src]


def _usage(src):
    _, remarks = _compile(src)
    res, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            res[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return res


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,patterns", [("gemm.hip", ["gemm_persistent_kernel", "gemm_pp_kernel", "gemm_wide_kernel", "gemm_kernelItNS0_4TileILi256ELi256", "gemm_resident_kernel"]),
                                          ("attention.hip", ["attn_mfma4_kernel", "attn_split_kernel"])])
def test_hot_kernels_use_no_scratch(src, patterns):
    res = _usage(src)
    # the scalar-epilogue fallback instantiations (VEC = false: N % 4 != 0) are not on the benchmark's path
    hot = {k: v for k, v in res.items() if any(p in k for p in patterns) and "ELb0ELb1ELb0EEEv" not in k}
    assert hot, f"no kernel of {src} matched {patterns}"
    for k, v in hot.items():
        # the fp8-WEIGHT instantiations of the round-2 persistent kernel (gemm_persistent_kernel<.., W8 = true>: precision fp8w and the
        # calibrating pass of precision fp8) keep a few tile-invariant address registers in scratch (stored once per launch, reloaded
        # once per tile, outside the K loop); they are not on the bench path (bf16 and fp8 run gemm_pp_kernel, which must be scratch-free)
        allowed = 32 if re.search(r"gemm_persistent_kernelILi\dELi\dELb1", k) else 0
        assert v.get("ScratchSize", 0) <= allowed, (k, v)
        assert v.get("VGPRs", 0) + v.get("AGPRs", 0) <= 256, (k, v)     # two waves per SIMD for the 512-thread GEMMs


def test_prefetched_epilogue_registers_are_not_touched_before_their_wait():
    """The persistent GEMM's epilogue prefetches its per-row operand (residual / gate) with inline-asm loads that hipcc does
    not track; the destination VGPRs count as written at the asm statement, so the compiler could legally read, copy or reuse
    them before the data lands (guide: cdna_hip_programming.md 5.7 item 1). This audits the generated ISA: between an asm
    `global_load_dwordx4` and the counted `s_waitcnt` that covers it, no other instruction may name those registers. Loads
    are issued up to three slabs ahead; the count of every wait says how many of the youngest stay in flight."""
    src, _ = _compile("gemm.hip")

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    checked = 0
    for m in re.finditer(r"^(_ZN4vima\S*gemm_(?:persistent|pp)_kernelILi\dELi([234])ELb[01]EEE\S*):", src, flags=re.M):
        body = src[m.end():src.index("s_endpgm", m.end())].split("\n")
        nit = 4 if m.group(2) == "3" else 2        # loads per slab: fp32 residual 4 row groups, gate / bf16 residual 2
        pending, in_asm, loads, since_wait = [], False, 0, 0
        for line in body:
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            toks = re.split(r"[\s,]+", t)
            if in_asm and toks[0] == "global_load_dwordx4":
                pending.append(regs(toks[1]))
                loads += 1
                since_wait += 1
                continue
            if in_asm and toks[0] == "s_waitcnt":
                # VMEM loads retire in order: a counted wait `vmcnt(n)` leaves the n youngest loads in flight (the slabs prefetched
                # ahead: up to three of them since round 6) and retires every older one
                mm = re.search(r"vmcnt\((\d+)\)", t)
                if mm:
                    n = int(mm.group(1))
                    if pending:     # an epilogue wait (the main loop's own counted waits find nothing of the epilogue pending)
                        assert n % nit == 0 and n <= 3 * nit, (m.group(1), t)
                    pending = pending[-n:] if n else []
                since_wait = 0
                continue
            if not in_asm and pending:
                inflight = set().union(*pending)
                used = set()
                for tk in toks[1:]:
                    used |= regs(tk)
                assert not (used & inflight), (m.group(1), t, sorted(used & inflight)[:4])
        # 16 (gate / bf16 residual) or 32 (fp32 residual) loads per epilogue; the epilogue is instantiated once per compile-time
        # (bias, row-scale) combination the kernel can take: 2 copies for these epilogues
        assert loads in (16, 32, 64), (m.group(1), loads)
        checked += 1
    assert checked >= 9, checked


def test_q4_kernel_stream_is_what_was_written():
    """gemm_q4_kernel (256x384, one wave per SIMD) owns its registers by hand: MFMAs and fragment reads are inline asm with tied operands, the
    lgkmcnt waits in front of the MFMAs are counted on the assumption that ONLY its own LDS reads are in flight, and its LDS-DMA helper writes M0
    without saving it. The ISA must therefore show, for every instantiation: no scratch; 256 AGPRs; and between the first and the last MFMA of the
    K loop no scalar load (SMEM returns out of order and shares lgkmcnt), no accumulator shuffle between the register files (v_accvgpr_*: the
    compiler cannot see the MFMA's latency), no spill traffic, no branch; M0 is named by the LDS-DMA helper only."""
    src, _ = _compile("gemm.hip")
    res = _usage("gemm.hip")
    q4 = {k: v for k, v in res.items() if "gemm_q4_kernel" in k}
    assert len(q4) >= 10, sorted(q4)                                   # five epilogue forms x two tiles (MIH = 2: 256x384, MIH = 1: 128x384)
    mih_of = lambda name: int(re.search(r"gemm_q4_kernelILi\d+ELi\d+ELi(\d)E", name).group(1))
    for k, v in q4.items():
        assert v.get("ScratchSize", 0) == 0, (k, v)
        # 16 (8) of the 24 (12) accumulator blocks live in AGPRs; the 128x384 tile's epilogue may park values in the free half of the AGPR file
        assert (v.get("AGPRs", 0) == 256 if mih_of(k) == 2 else 128 <= v.get("AGPRs", 0) <= 256) and v.get("VGPRs", 0) <= 256, (k, v)
    checked = 0
    for m in re.finditer(r"^(_ZN4vima\S*gemm_q4_kernel\S*):", src, flags=re.M):
        mih = mih_of(m.group(1))
        body = src[m.end():src.index(".Lfunc_end", m.end())].split("\n")      # (the kernel has an early exit: several s_endpgm)
        code = [ln.strip() for ln in body if ln.strip() and ln.strip()[0] not in ";."]
        idx = [i for i, ln in enumerate(code) if ln.startswith("v_mfma_f32_32x32x16_bf16")]
        assert len(idx) == 96 * mih, (m.group(1), len(idx))            # 8 phases x 24 (12): ONE loop body
        loop = code[idx[0]:idx[-1] + 1]
        bad = [ln for ln in loop if re.match(r"(s_load|s_buffer_load|v_accvgpr|scratch_|s_cbranch|v_readlane|v_writelane)", ln)]
        assert not bad, (m.group(1), bad[:4])
        # rolling reads per K-tile: P1 4 MIH, P2 12, P3 4 MIH, P4 4 (3 + MIH); the last two of the loop body follow its last MFMA
        assert sum(ln.startswith("ds_read_b128") for ln in loop) == 2 * (24 + 12 * mih) - 2, m.group(1)
        assert sum(ln.startswith("global_load_lds_dwordx4") for ln in loop) == 2 * (12 + 4 * mih), m.group(1)   # 2 K-tiles x (6 + 6 + 2 x 2 MIH) pieces
        assert sum(ln.startswith("s_barrier") for ln in loop) == 7, m.group(1)             # 8 phases (the last one's barrier follows the last MFMA)
        m0 = [ln for ln in code if re.search(r"\bm0\b", ln)]
        assert m0 and all(ln.startswith("s_mov_b32 m0, s") for ln in m0), (m.group(1), [ln for ln in m0 if not ln.startswith("s_mov_b32 m0, s")][:3])
        checked += 1
    assert checked >= 10, checked


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
def test_attention_prologue_issues_its_table_loads_in_batches():
    """attn_mfma4_kernel's mask and bias tables: hipcc once sank every table load into the branch that used it and waited for it there
    (6-8 serial memory latencies per workgroup, the first of them also waiting for the first key tile's LDS-DMA: -5 % on the T5 shape,
    profiles/r06_attention_experiments.txt). The source now issues four unconditional clamped loads per table pass and consumes them
    with an empty asm; this audits the generated ISA of the T5 instantiation: in front of the first barrier, every group of table
    loads is complete before the first wait that follows its first load, and the first tile's LDS-DMA is requested before any of them."""
    src, _ = _compile("attention.hip")
    m = re.search(r"^(_ZN4vima\S*attn_mfma4_kernelILi64ELi0ELi1EEE\S*):", src, flags=re.M)
    assert m, "attn_mfma4_kernel<64, T5, 1> not found"
    body = src[m.end():src.index("s_barrier", m.end())].split("\n")
    ops = [re.split(r"[\s,]+", l.strip())[0] + (" " + l.strip() if "s_waitcnt" in l else "") for l in body if l.strip() and l.strip()[0] not in ";."]
    first_dma = next(i for i, o in enumerate(ops) if o.startswith("global_load_lds_dwordx4"))
    for kind in ("global_load_ubyte", "global_load_dword"):
        idx = [i for i, o in enumerate(ops) if o.split(" ")[0] == kind]
        assert len(idx) >= 4, (kind, len(idx))
        assert first_dma < idx[0], "the first key tile is requested before the tables are read"
        # groups of four consecutive loads (no wait on the vector-memory counter between the first and the last of a group)
        for g in range(0, len(idx) - len(idx) % 4, 4):
            between = ops[idx[g]:idx[g + 3]]
            assert not any("vmcnt" in o for o in between), (kind, g, between)
