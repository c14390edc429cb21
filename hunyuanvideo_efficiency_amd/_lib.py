"""Bindings of the HIP extension.

* `libhv_kernels.so`   - the C ABI declared in include/hv_kernels.h (hipcc, gfx950).  `load()` binds it with ctypes: used by the
  symbol/ABI checks (tests/test_capi_cpu.py) and by any host that is not PyTorch.
* `libhv_torch_ops.so` - csrc/hv_torch_ops.cpp: TORCH_LIBRARY(hv, m) + TORCH_LIBRARY_IMPL(hv, CUDA, m), one custom op per C-ABI
  entry point (`torch.ops.hv.gemm_bf16`, `torch.ops.hv.attn_fwd_bf16`, ...), each launching on the current HIP stream of its
  tensors' device.  `call()` / `host()` are the product path: ops.py and vae_ops.py reach every kernel through torch.ops.hv.

There is NO fallback: if a shared library is missing or a symbol is absent the import of the compute path raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C hunyuanvideo_efficiency_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhv_kernels.so")
TORCH_OPS_PATH = os.path.join(_HERE, "lib", "libhv_torch_ops.so")
ABI_VERSION = 4

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype is always int); mirrors include/hv_kernels.h one to one
SIGNATURES = {
    "hv_abi_version": [],
    "hv_ln_modulate_bf16": [_p, _p, _p, _p, _l, _i, _l, _l, _f, _i, _p],
    "hv_qknorm_rope_bf16": [_p, _p, _p, _p, _p, _l, _l, _i, _i, _l, _l, _f, _p],
    "hv_qknorm_rope_scatter_bf16": [_p, _p, _p, _p, _p, _l, _l, _i, _i, _l, _l, _f, _p, _l, _i, _l, _p],
    "hv_gemm_bf16": [_p, _l, _p, _l, _p, _i, _i, _i, _p, _l, _i, _i, _p, _l, _i, _p, _p, _l, _p],
    "hv_linear_smallm_bf16": [_p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _i, _p],
    "hv_timestep_embedding_bf16": [_p, _p, _i, _i, _f, _p],
    "hv_attn_fwd_bf16": [_p, _p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _f, _p, _l, _p],
    "hv_attn_workspace_bytes": [_i, _i, _i],
    "hv_attn_w4_loop_signature": [],
    "hv_attn_partial_bf16": [_p, _p, _p, _l, _l, _l, _i, _i, _i, _i, _f, _p, _p, _i, _i, _i, _p],
    "hv_attn_merge_bf16": [_p, _p, _p, _l, _i, _i, _i, _p],
    "hv_attn_suggest_splits": [_i, _i, _i],
    "hv_patchify_f32_bf16": [_p, _p, _i, _i, _i, _i, _p],
    "hv_unpatchify_bf16": [_p, _p, _i, _i, _i, _i, _l, _p],
    "hv_euler_step_f32": [_p, _p, _f, _l, _p],
    "hv_euler_step_f32_f32": [_p, _p, _f, _l, _p],
    "hv_masked_mean_bf16": [_p, _p, _p, _i, _i, _p],
    "hv_broadcast_row_bf16": [_p, _p, _l, _i, _l, _p],
    "hv_copy3d_bf16": [_p, _p, _i, _l, _i, _l, _l, _l, _l, _p],
    "hv_fp8_dequant_bf16": [_p, _p, _p, _l, _p],
    "hv_ln_modulate_fp8": [_p, _p, _p, _p, _p, _l, _i, _l, _l, _f, _p],
    "hv_quant_rows_fp8": [_p, _l, _p, _l, _p, _l, _i, _p],
    "hv_gemm_fp8": [_p, _l, _p, _p, _l, _p, _p, _i, _i, _i, _p, _l, _i, _i, _p, _l, _i, _p, _p, _l, _p],
    "hv_gemm_f16": [_p, _l, _p, _l, _p, _i, _i, _i, _p, _l, _i, _p, _l, _p],
    "hv_conv3d_causal_f16": [_p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p],
    "hv_gn_partial_rows": [_l],
    "hv_subpixel_gn_partial_rows": [_i, _i, _i, _i],
    "hv_groupnorm_finalize_f16": [_p, _l, _l, _l, _i, _i, _f, _p, _p, _p, _p],
    "hv_groupnorm_affine_f16": [_p, _l, _l, _i, _i, _f, _p, _p, _p, _l, _p, _p],
    "hv_groupnorm_apply_f16": [_p, _l, _p, _l, _l, _i, _p, _i, _p],
    "hv_softmax_rows_f32_f16": [_p, _l, _p, _l, _i, _i, _i, _f, _i, _p],
    "hv_transpose_16b": [_p, _l, _p, _l, _i, _i, _p],
    "hv_conv3d_upsampled_subpixel_f16": [_p, _l, _p, _p, _i, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p, _l, _p],
    "hv_conv3d_cout4_f16": [_p, _l, _p, _i, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p, _l, _p],
    "hv_conv3d_cout4_planes_floats": [_l],
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
        fn.restype = C.c_int64 if name in ("hv_attn_workspace_bytes", "hv_gn_partial_rows", "hv_subpixel_gn_partial_rows", "hv_conv3d_cout4_planes_floats") else C.c_int
    v = lib.hv_abi_version()
    if v != ABI_VERSION:
        raise HVKernelError(f"libhv_kernels ABI {v} != expected {ABI_VERSION}: rebuild the extension")
    # the attention kernel's steady-state iteration is generated code (csrc/hv_attention_w4_loop.inc): a library compiled from another
    # iteration - a stale file or a timing experiment's, which computes garbage by design - must not pass for the product
    inc = os.path.join(_HERE, "csrc", "hv_attention_w4_loop.inc")
    if os.path.exists(inc) and os.environ.get("HV_ALLOW_EXPERIMENT_LIB") != "1":
        with open(inc) as f:
            m = re.search(r"#define HV_W4_LOOP_SIGNATURE 0x([0-9a-f]{8})u", f.read(600))
        if m is None or (lib.hv_attn_w4_loop_signature() & 0xFFFFFFFF) != int(m.group(1), 16):
            raise HVKernelError(f"{LIB_PATH} was not compiled from {inc}: rebuild the extension (make -C hunyuanvideo_efficiency_amd/csrc); "
                                "HV_ALLOW_EXPERIMENT_LIB=1 admits an experiment build (tools/attn_variants)")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise HVKernelError(f"{what} failed with code {code} "
                            f"({'bad argument' if code == -1 else 'launch failure' if code == -2 else 'unknown'})")


_hv = None


def torch_ops():
    """Load (once) libhv_torch_ops.so and return the `torch.ops.hv` namespace; raises HVKernelError if it is not built."""
    global _hv
    if _hv is not None:
        return _hv
    import torch
    if not os.path.exists(TORCH_OPS_PATH):
        raise HVKernelError(
            f"{TORCH_OPS_PATH} is missing: the PyTorch custom-op library is not built (run __graft_entry__.build()). "
            "hunyuanvideo_efficiency_amd has no CPU or eager fallback.")
    torch.ops.load_library(TORCH_OPS_PATH)
    hv = torch.ops.hv
    v = int(hv.abi_version())
    if v != ABI_VERSION:
        raise HVKernelError(f"libhv_torch_ops / libhv_kernels ABI {v} != expected {ABI_VERSION}: rebuild the extension")
    for name in SIGNATURES:
        if not hasattr(hv, name[len("hv_"):]):
            raise HVKernelError(f"{TORCH_OPS_PATH} does not register torch.ops.hv.{name[len('hv_'):]}")
    _hv = hv
    return hv


def call(name: str, *args):
    """torch.ops.hv.<name>(*args): tensors (or None), ints, floats, int lists - see the schema in csrc/hv_torch_ops.cpp.
    A non-zero kernel return code, a CPU tensor or a tensor on another GPU surfaces as HVKernelError."""
    op = getattr(torch_ops(), name)
    try:
        return op(*args)
    except (RuntimeError, NotImplementedError) as e:
        raise HVKernelError(f"hv::{name}: {str(e).splitlines()[0]}") from e


def host(name: str, *args) -> int:
    """Pure host queries of the ABI (no tensors): abi_version, attn_workspace_bytes, attn_suggest_splits."""
    return int(getattr(torch_ops(), name)(*args))
