"""CPU: the C-ABI library loads and exports exactly the symbols include/hv_kernels.h declares, and the
ctypes table mirrors the header (argument counts).  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "hv_kernels.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|int64_t)\s+(hv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return decls


def test_header_matches_ctypes_table():
    from hunyuanvideo_efficiency_amd import _lib
    decls = _header_decls()
    assert set(decls) == set(_lib.SIGNATURES), (set(decls) ^ set(_lib.SIGNATURES))
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name]) == n, name


def test_library_loads_and_exports_every_symbol():
    from hunyuanvideo_efficiency_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} not built: run __graft_entry__.build()")
    lib = _lib.load()
    for name in _header_decls():
        assert hasattr(lib, name), name
    assert lib.hv_abi_version() == _lib.ABI_VERSION


def test_ops_refuse_cpu_tensors():
    import torch
    from hunyuanvideo_efficiency_amd import ops, _lib
    with pytest.raises(_lib.HVKernelError):
        ops.ln_modulate(torch.zeros(4, 256, dtype=torch.bfloat16))
    with pytest.raises(_lib.HVKernelError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))


def test_torch_custom_ops_registered_for_every_entry_point():
    """SURVEY.md 8b last row: every C-ABI entry point is a PyTorch custom op torch.ops.hv.<name> (TORCH_LIBRARY(hv, m)); the
    kernels are registered for the CUDA(=HIP) dispatch key only - a CPU tensor is refused by the dispatcher (no CPU path)."""
    import torch
    from hunyuanvideo_efficiency_amd import _lib
    hv = _lib.torch_ops()
    for name in _header_decls():
        op = getattr(hv, name[len("hv_"):])
        assert op.default._schema.name == "hv::" + name[len("hv_"):]
    assert int(hv.abi_version()) == _lib.ABI_VERSION
    assert _lib.host("attn_suggest_splits", 118811, 118811, 3) == 2        # Ulysses-8 shape: shallow grid -> KV split
    assert _lib.host("attn_suggest_splits", 118811, 118811, 24) == 1
    # rows of the GroupNorm-statistics buffer a conv epilogue fills: one per 64 output rows, whole 256-row tiles; the sub-pixel
    # upsampler forms its tiles per output parity class (8 classes when T is upsampled: 4 with sT frames, 4 with sT - 1)
    assert _lib.host("gn_partial_rows", 65 * 256 * 256) == 66560 and _lib.host("gn_partial_rows", 90) == 4
    assert _lib.host("subpixel_gn_partial_rows", 17, 64, 64, 1) == 4 * (4 * 272 + 4 * 256)
    assert _lib.host("subpixel_gn_partial_rows", 1, 4, 4, 1) == 4 * 4 and _lib.host("subpixel_gn_partial_rows", 65, 128, 128, 0) == 4 * 4 * 4160
    with pytest.raises(_lib.HVKernelError):
        _lib.call("euler_step_f32", torch.zeros(4), torch.zeros(4, dtype=torch.bfloat16), 0.1, 4)


def test_generated_torch_ops_source_is_up_to_date():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_torch_ops", os.path.join(ROOT, "tools", "gen_torch_ops.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert open(g.OUT).read() == g.gen(), "include/hv_kernels.h changed: run python tools/gen_torch_ops.py"
