"""ctypes binding of libhv_kernels.so (the C ABI declared in include/hv_kernels.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import of the
compute path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C hunyuanvideo_efficiency_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhv_kernels.so")
ABI_VERSION = 1

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype is always int); mirrors include/hv_kernels.h one to one
SIGNATURES = {
    "hv_abi_version": [],
    "hv_ln_modulate_bf16": [_p, _p, _p, _p, _l, _i, _l, _l, _f, _i, _p],
    "hv_qknorm_rope_bf16": [_p, _p, _p, _p, _p, _l, _l, _i, _i, _l, _l, _f, _p],
    "hv_gemm_bf16": [_p, _l, _p, _l, _p, _i, _i, _i, _p, _l, _i, _i, _p, _l, _i, _p, _p, _l, _p],
    "hv_linear_smallm_bf16": [_p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _i, _p],
    "hv_timestep_embedding_bf16": [_p, _p, _i, _i, _f, _p],
    "hv_attn_fwd_bf16": [_p, _p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _f, _p, _l, _p],
    "hv_attn_workspace_bytes": [_i, _i, _i],
    "hv_attn_partial_bf16": [_p, _p, _p, _l, _l, _l, _i, _i, _i, _i, _f, _p, _p, _i, _i, _i, _p],
    "hv_attn_merge_bf16": [_p, _p, _p, _l, _i, _i, _i, _p],
    "hv_attn_suggest_splits": [_i, _i, _i],
    "hv_patchify_f32_bf16": [_p, _p, _i, _i, _i, _i, _p],
    "hv_unpatchify_bf16": [_p, _p, _i, _i, _i, _i, _l, _p],
    "hv_euler_step_f32": [_p, _p, _f, _l, _p],
    "hv_masked_mean_bf16": [_p, _p, _p, _i, _i, _p],
    "hv_broadcast_row_bf16": [_p, _p, _l, _i, _l, _p],
    "hv_copy3d_bf16": [_p, _p, _i, _l, _i, _l, _l, _l, _l, _p],
    "hv_fp8_dequant_bf16": [_p, _p, _p, _l, _p],
    "hv_gemm_f16": [_p, _l, _p, _l, _p, _i, _i, _i, _p, _l, _i, _p, _l, _p],
    "hv_conv3d_causal_f16": [_p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p],
    "hv_groupnorm_affine_f16": [_p, _l, _l, _i, _i, _f, _p, _p, _p, _l, _p, _p],
    "hv_groupnorm_apply_f16": [_p, _l, _p, _l, _l, _i, _p, _i, _p],
    "hv_softmax_rows_f32_f16": [_p, _l, _p, _l, _i, _i, _i, _f, _p],
    "hv_transpose_16b": [_p, _l, _p, _l, _i, _i, _p],
    "hv_conv3d_causal_strided_f16": [_p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hv_temporal_resample_f16": [_p, _l, _p, _l, _i, _l, _i, _i, _i, _i, _p],
    "hv_vae_latent_tile_f16": [_p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _p, _p],
    "hv_vae_blend_f16": [_p, _p, _p, _p, _p, _i, _i, _p],
    "hv_copy4d_16b": [_p, _p, _p, _p, _p, _p],
    "hv_vae_postprocess_f16_f32": [_p, _p, _l, _p],
}


class HVKernelError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises HVKernelError if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HVKernelError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()). "
            "hunyuanvideo_efficiency_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HVKernelError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name == "hv_attn_workspace_bytes" else C.c_int
    v = lib.hv_abi_version()
    if v != ABI_VERSION:
        raise HVKernelError(f"libhv_kernels ABI {v} != expected {ABI_VERSION}: rebuild the extension")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise HVKernelError(f"{what} failed with code {code} "
                            f"({'bad argument' if code == -1 else 'launch failure' if code == -2 else 'unknown'})")
