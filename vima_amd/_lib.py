"""ctypes binding of libvima_hip.so (C ABI declared in include/vima_hip.h).

The HIP library is the product: there is no CPU / PyTorch fallback. If the shared object has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `bash vima_amd/csrc/build.sh`) importing this module
raises immediately.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIMA_HIP_LIB: experiment builds only (scripts/build_ablate.sh A/B runs); the product library is the in-tree one
LIB_PATH = os.environ.get("VIMA_HIP_LIB") or os.path.join(_HERE, "lib", "libvima_hip.so")

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
vp = ctypes.c_void_p


class VimaConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("embed_dim", "xf_n_layers", "sattn_n_heads", "xattn_n_heads",
                                     "xattn_n_positions", "n_positions", "precision", "policy_kind")]


def _header_abi_version() -> int:
    """VIMA_ABI_VERSION the package was built against. The number lives in include/vima_hip.h; `vima_amd/csrc/build.sh` copies it into
    vima_amd/_abi.py next to the library, so that a deployment which ships only the package directory (+ the .so, or a VIMA_HIP_LIB
    override) still imports. In the repository the header is read and must agree with the baked copy (tests/test_abi.py)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "vima_hip.h")
    if os.path.exists(hdr):
        with open(hdr) as f:
            m = re.search(r"^#define\s+VIMA_ABI_VERSION\s+(\d+)", f.read(), flags=re.M)
        if not m:
            raise RuntimeError(f"VIMA_ABI_VERSION not found in {hdr}")
        return int(m.group(1))
    try:
        from ._abi import ABI_VERSION as baked
        return int(baked)
    except Exception:   # noqa: BLE001 -- neither the header nor the baked copy: the version check in load() is skipped
        import warnings
        warnings.warn("vima_amd: neither include/vima_hip.h nor vima_amd/_abi.py found; the library's ABI version is not checked")
        return -1


ABI_VERSION = _header_abi_version()
PRECISION = {"fp32": 0, "bf16": 1, "fp8w": 2, "fp8": 3}
POLICY_KIND = {"vima": 0, "gpt": 1, "gato": 2, "flamingo": 3}   # VIMA_POLICY_* (include/vima_hip.h)

# exported symbol -> (restype, argtypes); must list every function declared in include/vima_hip.h
PROTOTYPES = {
    "vima_create": (ctypes.c_int, [ctypes.POINTER(VimaConfig), ctypes.c_int, ctypes.POINTER(vp)]),
    "vima_destroy": (None, [vp]),
    "vima_last_error": (ctypes.c_char_p, []),
    "vima_abi_version": (ctypes.c_int, []),
    "vima_set_param": (ctypes.c_int, [vp, ctypes.c_char_p, vp, ctypes.POINTER(c_i64), ctypes.c_int]),
    "vima_finalize_params": (ctypes.c_int, [vp]),
    "vima_required_params": (c_i64, [ctypes.POINTER(VimaConfig), ctypes.c_char_p, c_i64]),
    "vima_obj_encode": (ctypes.c_int, [vp, vp * 2, vp * 2, ctypes.c_int, ctypes.c_int, vp, vp]),
    "vima_obs_encode": (ctypes.c_int, [vp, vp * 2, vp * 2, vp * 2, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "vima_prompt_encode": (ctypes.c_int, [vp, vp, ctypes.c_int, vp * 2, vp * 2, vp * 2, ctypes.c_int, ctypes.c_int,
                                          vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "vima_t5_encode": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "vima_decode": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   vp, c_i64, c_i64, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "vima_decode_step": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, c_i64, c_i64, vp,
                                        ctypes.c_int, vp, vp]),
    "vima_decode_restart": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, c_i64, c_i64, vp, ctypes.c_int, vp]),
    "vima_rgb_tokens_per_image": (ctypes.c_int, [ctypes.POINTER(VimaConfig)]),
    "vima_rgb_encode": (ctypes.c_int, [vp, vp * 2, ctypes.c_int, vp, vp]),
    "vima_rgb_obs_encode": (ctypes.c_int, [vp, vp * 2, vp, ctypes.c_int, vp, vp]),
    "vima_rgb_prompt_encode": (ctypes.c_int, [vp, vp, ctypes.c_int, vp * 2, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "vima_seq_decode": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, c_i64, c_i64, vp, ctypes.c_int,
                                       vp, vp]),
    "vima_action_head": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp]),
    "vima_action_embed": (ctypes.c_int, [vp, vp * 4, ctypes.c_int, vp, vp]),
    "vima_op_linear": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      vp, vp]),
    "vima_op_layernorm": (ctypes.c_int, [vp, vp, vp, vp, c_f32, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "vima_op_attention": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, c_f32, ctypes.c_int, ctypes.c_int, vp, vp]),
    "vima_t5_bucket": (ctypes.c_int, [ctypes.c_int]),
    "vima_fp8_e4m3_encode": (None, [vp, vp, c_i64]),
    "vima_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, c_i64]),
    "vima_prof_enable": (ctypes.c_int, [vp, ctypes.c_int]),
    "vima_prof_read": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                                      ctypes.POINTER(ctypes.c_double)]),
    "vima_prof_read_ex": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "vima_prof_read_gemm_kernels": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double),
                                                   ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "vima_prof_read_gemm_launches": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                                    ctypes.POINTER(ctypes.c_float)]),
    "vima_fp8_act_scales": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    "vima_workspace_bytes": (c_i64, [vp]),
    "vima_graph_stats": (ctypes.c_int, [vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "vima_crop_objects": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp,
                                         vp, vp]),
    "vima_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "vima_comm_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]),
    "vima_comm_world": (ctypes.c_int, [vp]),
    "vima_comm_rank": (ctypes.c_int, [vp]),
    "vima_allgather_logits": (ctypes.c_int, [vp, vp, vp, c_i64, ctypes.c_int, vp]),
    "vima_comm_destroy": (None, [vp]),
}
COMM_ID_BYTES = 128

_lib = None


def load():
    """Load (once) and return the ctypes handle of libvima_hip.so; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP library is required (no CPU fallback). "
            "Build it with `bash vima_amd/csrc/build.sh` (hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if ABI_VERSION >= 0 and lib.vima_abi_version() != ABI_VERSION:   # a stale build of the library against a newer header (or the reverse)
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.vima_abi_version()}, include/vima_hip.h declares {ABI_VERSION}: rebuild "
                           "with `bash vima_amd/csrc/build.sh`")
    _lib = lib
    return lib


class VimaError(RuntimeError):
    pass


# error codes the C side uses for the reference's exception types
_EXC = {22: ValueError, 33: AssertionError, 34: IndexError}


def check(code: int):
    if code != 0:
        msg = load().vima_last_error().decode("utf-8", "replace")
        raise _EXC.get(code, VimaError)(msg)


def required_params(cfg: VimaConfig) -> list[str]:
    lib = load()
    n = lib.vima_required_params(ctypes.byref(cfg), None, 0)
    buf = ctypes.create_string_buffer(int(n) + 1)
    lib.vima_required_params(ctypes.byref(cfg), buf, n)
    return [k.decode() for k in buf.raw[:n].split(b"\0") if k]
