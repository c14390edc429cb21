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


def test_generated_attention_iteration_is_up_to_date_and_consistent():
    """hv_attention_w4_loop.inc (the steady-state attention iteration as one asm statement) == what tools/gen_attn_w4_asm.py emits,
    and the schedule tables it is built from are self-consistent: every exponential is issued exactly once, every packed P word is
    written before the P.V MFMA that reads it (with at least one gap in between: VALU write -> MFMA operand wait states), every
    fragment read precedes its first MFMA by PF fragments and the counted lgkmcnt of that MFMA equals the LDS instructions issued
    in between."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_attn_w4_asm", os.path.join(ROOT, "tools", "gen_attn_w4_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert not g.STAMPS and not g.ABL
    assert open(g.OUT).read() == g.main(), "run python tools/gen_attn_w4_asm.py"
    # ... and the built library was compiled from exactly this file (not from a stale one, not from a timing experiment's)
    import ctypes
    from hunyuanvideo_efficiency_amd import _lib
    sig = int(re.search(r"#define HV_W4_LOOP_SIGNATURE 0x([0-9a-f]{8})u", open(g.OUT).read()).group(1), 16)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert (lib.hv_attn_w4_loop_signature() & 0xFFFFFFFF) == sig, "libhv_kernels.so is not built from the in-tree iteration: make -C hunyuanvideo_efficiency_amd/csrc"
    body = g.gen_iter(0)
    gaps, cur = [], None
    for ln in body:
        if ln.startswith("; ---- gap"):
            cur = []
            gaps.append(cur)
        else:
            cur.append(ln)
    assert len(gaps) == 64 and all(sum("v_mfma" in x for x in gp) == 1 for gp in gaps)
    exps = [ln for gp in gaps for ln in gp if ln.startswith("v_exp_f32")]
    assert len(exps) == 64 and len({ln.split(",")[1].strip() for ln in exps}) == 64            # every S'(t) register exactly once
    assert sum(ln.startswith("v_add_f32") for gp in gaps for ln in gp) == 64
    assert sum(ln.startswith("v_cvt_pk") for gp in gaps for ln in gp) == 32
    assert sum(ln.startswith("buffer_load") for gp in gaps for ln in gp) == 8
    # packed P words: written (v_cvt_pk dest) in a gap strictly before the MFMA that reads their tuple
    written = {}
    for gi, gp in enumerate(gaps):
        for ln in gp:
            if ln.startswith("v_cvt_pk"):
                written[int(re.match(r"v_cvt_pk_bf16_f32 v(\d+),", ln).group(1))] = gi
    for gi, gp in enumerate(gaps[32:], start=32):
        m = re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], a\[\d+:\d+\], v\[(\d+):(\d+)\]", [x for x in gp if "v_mfma" in x][0])
        lo, hi = int(m.group(1)), int(m.group(2))
        assert all(written[r] < gi for r in range(lo, hi + 1)), (gi, lo, hi)
    # fragment reads / counted waits: replay the LDS queue
    issued = []          # (fragment slot register, gap) in issue order
    for gi, gp in enumerate(gaps):
        for ln in gp:
            if ln.startswith("ds_read"):
                issued.append((int(re.search(r"a\[(\d+):", ln).group(1)), gi))
    assert len(issued) == 16 + 32
    # steady state: the stream of the previous iteration precedes this one; a wait lgkmcnt(N) at gap g leaves the N youngest of the reads
    # issued before it outstanding - the fragment registers of this gap's MFMA (and of the next fragment) must not be among them
    prev = [(r, gi - 64) for r, gi in issued]
    for gi, gp in enumerate(gaps):
        w = [x for x in gp if x.startswith("s_waitcnt lgkmcnt")]
        if not w:
            continue
        n = int(re.search(r"lgkmcnt\((\d+)\)", w[0]).group(1))
        before = [x for x in prev + issued if x[1] < gi]
        pending = {r for r, _ in before[len(before) - n:]} if n else set()
        mf = [x for x in gp if "v_mfma" in x][0]
        frag = int(re.findall(r"a\[(\d+):\d+\]", mf)[1 if gi >= 32 else 0])
        assert frag >= 192 and frag not in pending and not any(r in (frag, frag + 2) for r in pending), (gi, n, frag, pending)
        # one wait per PAIR of fragments: the MFMAs two gaps later (next fragment) issue without a wait of their own
        if gi + 2 < 64:
            mf2 = [x for x in gaps[gi + 2] if "v_mfma" in x][0]
            assert not [x for x in gaps[gi + 2] if x.startswith("s_waitcnt lgkmcnt")]
            frag2 = int(re.findall(r"a\[(\d+):\d+\]", mf2)[1 if gi + 2 >= 32 else 0])
            assert not any(r in (frag2, frag2 + 2) for r in pending), (gi, n, frag2, pending)


def test_attention_kernel_generated_code_audit(tmp_path):
    """hv_attention_w4.hip owns the accumulator half of the register file by literal names, so what hipcc does around the asm matters:
    compile it for gfx950 with -save-temps and require (cdna_hip_programming.md 5.7 item 4) no VGPR spills, no scratch, all 256
    accumulator registers allocated, and not one compiler-generated v_accvgpr_* outside ASMSTART/ASMEND (hipcc parks long-lived
    values in the accumulator half when it is short of registers - that would be a silent write into O)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "hunyuanvideo_efficiency_amd", "csrc", "hv_attention_w4.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-save-temps=obj", "-c", src,
                    "-o", str(tmp_path / "w4.o")], check=True, cwd=str(tmp_path), capture_output=True)
    asm = open(tmp_path / "hv_attention_w4-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    meta = {k: int(v) for k, v in re.findall(r"\.(vgpr_spill_count|private_segment_fixed_size|agpr_count|sgpr_spill_count):\s+(\d+)", asm)}
    assert meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0 and meta["agpr_count"] == 256, meta
    inside, bad = False, []
    for i, ln in enumerate(asm.split("\n")):
        if "#ASMSTART" in ln:
            inside = True
        elif "#ASMEND" in ln:
            inside = False
        elif not inside and "v_accvgpr" in ln:
            bad.append((i, ln.strip()))
    assert not bad, bad[:5]
    assert "scratch_" not in asm
