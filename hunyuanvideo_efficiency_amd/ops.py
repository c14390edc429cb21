"""Tensor-level wrappers over the PyTorch custom ops `torch.ops.hv.*` (csrc/hv_torch_ops.cpp: one op per C-ABI entry point of
include/hv_kernels.h, registered with TORCH_LIBRARY(hv, m) for the CUDA(=HIP) dispatch key).  torch supplies device memory and
the current HIP stream only; every arithmetic operation is a hand-written gfx950 kernel.  All wrappers raise if given a
non-GPU tensor: there is no CPU path in the product."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

BF16 = torch.bfloat16
ACT_NONE, ACT_GELU_TANH, ACT_SILU = 0, 1, 2
PROFILE_ATTN = None  # set to a list by bench.py to collect (start_event, end_event, n_q, n_kv, heads) per launch


def _chk(t: torch.Tensor, dtype, name: str, last_contig: bool = True):
    if not t.is_cuda:
        raise _lib.HVKernelError(f"{name}: expected a GPU tensor (this package has no CPU path), got {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if last_contig and t.dim() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def _rows(t: torch.Tensor, name: str):
    """view a [.., D] tensor as rows: returns (n_rows, D, row_stride); leading dims must be collapsible"""
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    d = t.shape[-1]
    n = t.numel() // d
    ld = t.stride(-2)
    # every leading dim must be a multiple walk of ld
    exp = ld
    for i in range(t.dim() - 2, -1, -1):
        if t.shape[i] != 1 and t.stride(i) != exp:
            raise ValueError(f"{name}: rows are not uniformly strided {t.shape} {t.stride()}")
        exp *= t.shape[i]
    return n, d, ld


def ln_modulate(x, shift=None, scale=None, out=None, eps: float = 1e-6, affine: bool = False):
    """mode 0: LN(x)*bf16(1+scale)+shift ; affine=True: LN(x)*scale(weight)+shift(bias)."""
    _chk(x, BF16, "x")
    m, d, ldx = _rows(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _chk(out, BF16, "out")
    mo, do, ldo = _rows(out, "out")
    assert (mo, do) == (m, d)
    for t, n in ((shift, "shift"), (scale, "scale")):
        if t is not None:
            _chk(t, BF16, n)
            assert t.numel() == d and t.is_contiguous(), f"{n} must be a contiguous [{d}] vector (batch 1)"
    _lib.call("ln_modulate_bf16", x, shift, scale, out, m, d, ldx, ldo, eps,
                                               1 if affine else 0)
    return out


def qknorm_rope_(qkv, q_weight, k_weight, cos, sin, n_rope: int, n_heads: int, k_offset: int, eps: float = 1e-6, out=None):
    """in place on qkv rows [n_rows, ld]: RMSNorm(q), RMSNorm(k) per head, RoPE on rows [0, n_rope).  With `out` (a 3-D view
    [n_rows, blocks, heads_per_block*128]): out of place, the 2*n_heads heads scattered by head block (the Ulysses send layout)."""
    _chk(qkv, BF16, "qkv")
    n, _, ld = _rows(qkv, "qkv")
    _chk(q_weight, BF16, "q_weight"), _chk(k_weight, BF16, "k_weight")
    if n_rope > 0:
        _chk(cos, torch.float32, "cos"), _chk(sin, torch.float32, "sin")
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == 128 and cos.shape[0] >= n_rope
    if out is not None:
        # out: [n_rows, blocks, heads_per_block*128] view (any block stride): head block b of row r at out[r, b, :]
        _chk(out, BF16, "out", False)
        assert out.dim() == 3 and out.shape[0] == n and out.stride(2) == 1 and out.shape[2] % 128 == 0
        assert out.shape[1] * out.shape[2] == 2 * n_heads * 128, (out.shape, n_heads)
        _lib.call("qknorm_rope_scatter_bf16", qkv, q_weight, k_weight, cos, sin, n, n_rope, n_heads, 128, ld, k_offset, eps,
                  out, out.stride(0), out.shape[2] // 128, out.stride(1))
        return out
    _lib.call("qknorm_rope_bf16", qkv, q_weight, k_weight, cos, sin, n,
                                               n_rope, n_heads, 128, ld, k_offset, eps)
    return qkv


def gemm(a, w, bias=None, out=None, act: int = ACT_NONE, n_split: int = 0, out1=None, act1: int = ACT_NONE,
         gate=None, res=None):
    """out = a @ w.T + bias with the fused epilogues of hv_gemm_bf16.  a:[M,K] w:[N,K] (row strides free)."""
    _chk(a, BF16, "a"), _chk(w, BF16, "w")
    m, k, lda = _rows(a, "a")
    n, kw, ldw = _rows(w, "w")
    assert k == kw, (a.shape, w.shape)
    n0 = n_split if 0 < n_split < n else n
    if out is None:
        out = torch.empty(*a.shape[:-1], n0, dtype=BF16, device=a.device)
    _chk(out, BF16, "out")
    mo, no, ld0 = _rows(out, "out")
    assert mo == m and no >= n0
    ld1 = 0
    if n0 < n:
        _chk(out1, BF16, "out1")
        m1, n1, ld1 = _rows(out1, "out1")
        assert m1 == m and n1 >= n - n0
    for t, nm in ((bias, "bias"), (gate, "gate")):
        if t is not None:
            _chk(t, BF16, nm)
            assert t.numel() == n and t.is_contiguous()
    ld_res = 0
    if res is not None:
        _chk(res, BF16, "res")
        mr, nr, ld_res = _rows(res, "res")
        assert mr == m and nr == n
    _lib.call("gemm_bf16", a, lda, w, ldw, bias, m, n, k, out, ld0, act, n0,
                                        out1, ld1, act1, gate, res, ld_res)
    return out


def linear_smallm(x, w, bias=None, silu_in: bool = False, silu_out: bool = False, out=None, addend=None):
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    m, k, ldx = _rows(x, "x")
    n, kw, ldw = _rows(w, "w")
    assert k == kw and ldw == k and m <= 4
    if out is None:
        out = torch.empty(*x.shape[:-1], n, dtype=BF16, device=x.device)
    _, _, ldo = _rows(out, "out")
    if bias is not None:
        _chk(bias, BF16, "bias")
    if addend is not None:
        _chk(addend, BF16, "addend")
        assert addend.shape == out.shape and addend.stride() == out.stride()
    _lib.call("linear_smallm_bf16", x, w, bias, addend, out, m, n, k, ldx, ldo,
                                                 (1 if silu_in else 0) | (2 if silu_out else 0))
    return out


_attn_ws = {}


ATTN_MIN_WORKSPACE = 256      # hv_attention.hpp KMAX_BYTES: the per-head key-norm bound alone (static-maximum mode of the kernel)


def _attn_workspace(n_q, n_kv, n_heads, device):
    """Scratch of hv_attn_fwd_bf16: 256 bytes for the per-head key-norm bound (always, when the key range is long enough for the
    pre-pass to pay) + the partials of the optional KV split (load balance; only allocated for shallow grids)."""
    if n_kv < 64 * 64:
        return None
    n_wg = ((n_q + 255) // 256) * n_heads
    need = ATTN_MIN_WORKSPACE if n_wg >= 16 * 256 else int(_lib.host("attn_workspace_bytes", n_q, n_kv, n_heads))
    key = str(device)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _attn_ws[key] = ws
    return ws


def attn_fwd(q, k, v, out, n_heads: int, scale: Optional[float] = None, kv_split_workspace: bool = True):
    """q:[n_q, >=H*128] k,v:[n_kv, ...] out:[n_q, ...] 2-D views (token rows, head h at column h*128)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        _chk(t, BF16, nm)
        assert t.dim() == 2
    n_q, n_kv = q.shape[0], k.shape[0]
    assert v.shape[0] == n_kv and out.shape[0] == n_q
    if scale is None:
        scale = 128 ** -0.5
    prof = PROFILE_ATTN
    if prof is not None:   # bench.py: HIP events on the launch stream around the dominant kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = _attn_workspace(n_q, n_kv, n_heads, q.device) if kv_split_workspace else None
    _lib.call("attn_fwd_bf16", q, k, v, out, q.stride(0), k.stride(0), v.stride(0),
                                            out.stride(0), n_q, n_kv, n_heads, 128, scale, ws,
                                            0 if ws is None else ws.numel())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, n_q, n_kv, n_heads))
    return out


class AttnPartials:
    """Slots of unnormalised attention partials (ring attention): part_o fp32 [n_slots, n_q, H, 128], part_ml [n_slots, n_q, H, 2]."""

    def __init__(self, n_slots: int, n_q: int, n_heads: int, device):
        self.n_slots, self.n_q, self.n_heads = n_slots, n_q, n_heads
        self.o = torch.empty(n_slots, n_q, n_heads, 128, dtype=torch.float32, device=device)
        self.ml = torch.empty(n_slots, n_q, n_heads, 2, dtype=torch.float32, device=device)
        self.used = 0


def attn_suggest_splits(n_q: int, n_kv: int, n_heads: int) -> int:
    return int(_lib.host("attn_suggest_splits", n_q, n_kv, n_heads))


def attn_partial(q, k, v, parts: AttnPartials, n_heads: int, splits: int = 1, scale: Optional[float] = None):
    """Attention of q against ONE K/V chunk; appends `splits` slots to `parts` (hv_attn_partial_bf16)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, BF16, nm)
        assert t.dim() == 2
    n_q, n_kv = q.shape[0], k.shape[0]
    assert v.shape[0] == n_kv and n_q == parts.n_q and n_heads == parts.n_heads and parts.used + splits <= parts.n_slots
    if scale is None:
        scale = 128 ** -0.5
    prof = PROFILE_ATTN
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("attn_partial_bf16", q, k, v, q.stride(0), k.stride(0), v.stride(0), n_q, n_kv, n_heads,
                                                128, scale, parts.o, parts.ml, parts.n_slots, parts.used, splits)
    if prof is not None:
        e1.record()
        prof.append((e0, e1, n_q, n_kv, n_heads))
    parts.used += splits


def attn_merge(parts: AttnPartials, out):
    """Fold the used slots into normalised bf16 rows out[n_q, >= H*128] (hv_attn_merge_bf16)."""
    _chk(out, BF16, "out")
    assert out.dim() == 2 and out.shape[0] == parts.n_q and parts.used >= 1
    _lib.call("attn_merge_bf16", parts.o, parts.ml, out, out.stride(0), parts.n_q, parts.n_heads,
                                              parts.used)
    return out


def patchify(x_f32, out=None):
    """x:[C,T,H,W] fp32 -> [T*(H/2)*(W/2), C*4] bf16"""
    _chk(x_f32, torch.float32, "x")
    assert x_f32.is_contiguous() and x_f32.dim() == 4
    c, t, h, w = x_f32.shape
    if out is None:
        out = torch.empty(t * (h // 2) * (w // 2), c * 4, dtype=BF16, device=x_f32.device)
    _lib.call("patchify_f32_bf16", x_f32, out, c, t, h, w)
    return out


def unpatchify(y, c: int, t: int, h: int, w: int, out=None):
    _chk(y, BF16, "y")
    assert y.dim() == 2
    if out is None:
        out = torch.empty(c, t, h, w, dtype=BF16, device=y.device)
    _lib.call("unpatchify_bf16", y, out, c, t, h, w, y.stride(0))
    return out


def euler_step_(sample_f32, model_out_bf16, dt: float):
    _chk(sample_f32, torch.float32, "sample"), _chk(model_out_bf16, BF16, "model_out")
    assert sample_f32.is_contiguous() and model_out_bf16.is_contiguous() and sample_f32.numel() == model_out_bf16.numel()
    _lib.call("euler_step_f32", sample_f32, model_out_bf16, float(dt), sample_f32.numel())
    return sample_f32


def euler_step_f32_(sample_f32, model_out_f32, dt: float):
    _chk(sample_f32, torch.float32, "sample"), _chk(model_out_f32, torch.float32, "model_out")
    assert sample_f32.is_contiguous() and model_out_f32.is_contiguous() and sample_f32.numel() == model_out_f32.numel()
    _lib.call("euler_step_f32_f32", sample_f32, model_out_f32, float(dt), sample_f32.numel())
    return sample_f32


def masked_mean(x, mask_i32=None):
    _chk(x, BF16, "x")
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty(x.shape[1], dtype=BF16, device=x.device)
    if mask_i32 is not None:
        _chk(mask_i32, torch.int32, "mask")
    _lib.call("masked_mean_bf16", x, mask_i32, out, x.shape[0], x.shape[1])
    return out


def broadcast_row_(src, dst):
    _chk(src, BF16, "src"), _chk(dst, BF16, "dst")
    assert dst.dim() == 2 and src.numel() == dst.shape[1]
    _lib.call("broadcast_row_bf16", src, dst, dst.shape[0], dst.shape[1], dst.stride(0))
    return dst


def timestep_embedding(t_f32, dim: int = 256, max_period: float = 10000.0):
    _chk(t_f32, torch.float32, "t")
    t_f32 = t_f32.reshape(-1).contiguous()
    out = torch.empty(t_f32.numel(), dim, dtype=BF16, device=t_f32.device)
    _lib.call("timestep_embedding_bf16", t_f32, out, t_f32.numel(), dim, max_period)
    return out


def fp8_dequant(w8, scale_bf16, out_bf16):
    """out = bf16(bf16(w8) * scale): w8 float8_e4m3fn (any shape, contiguous), scale a 1-element bf16 tensor."""
    if not w8.is_cuda or w8.dtype != torch.float8_e4m3fn:
        raise _lib.HVKernelError("fp8_dequant: expected a float8_e4m3fn GPU tensor")
    _chk(scale_bf16, BF16, "scale"), _chk(out_bf16, BF16, "out")
    assert w8.is_contiguous() and out_bf16.is_contiguous() and out_bf16.numel() == w8.numel()
    _lib.call("fp8_dequant_bf16", w8, scale_bf16, out_bf16, w8.numel())
    return out_bf16


# ---------------------------------------------------------------------------------------------- FP8-MFMA path (opt-in)
FP8 = torch.float8_e4m3fn


def ln_modulate_fp8(x, shift=None, scale=None, out_q=None, out_scale=None, eps: float = 1e-6):
    """bf16(LN(x)*bf16(1+scale)+shift) quantised per row -> (e4m3 rows [M, D], fp32 row scales [M])."""
    _chk(x, BF16, "x")
    m, d, ldx = _rows(x, "x")
    if out_q is None:
        out_q = torch.empty(m, d, dtype=FP8, device=x.device)
    if out_scale is None:
        out_scale = torch.empty(m, dtype=torch.float32, device=x.device)
    assert out_q.dtype == FP8 and out_q.is_cuda and out_q.shape[-1] == d and out_q.stride(-1) == 1 and out_scale.numel() >= m
    for t, n in ((shift, "shift"), (scale, "scale")):
        if t is not None:
            _chk(t, BF16, n)
            assert t.numel() == d and t.is_contiguous()
    _lib.call("ln_modulate_fp8", x, shift, scale, out_q, out_scale, m, d, ldx, out_q.stride(-2) if out_q.dim() > 1 else d, eps)
    return out_q, out_scale


def quant_rows_fp8(x, out_q=None, out_scale=None):
    """bf16 rows [M, K] -> (e4m3 rows, fp32 row scales): per-token dynamic quantisation of a GEMM A operand."""
    _chk(x, BF16, "x")
    m, k, ldx = _rows(x, "x")
    if out_q is None:
        out_q = torch.empty(m, k, dtype=FP8, device=x.device)
    if out_scale is None:
        out_scale = torch.empty(m, dtype=torch.float32, device=x.device)
    assert out_q.dtype == FP8 and out_q.is_cuda and out_q.shape[-1] == k and out_q.stride(-1) == 1 and out_scale.numel() >= m
    _lib.call("quant_rows_fp8", x, ldx, out_q, out_q.stride(-2) if out_q.dim() > 1 else k, out_scale, m, k)
    return out_q, out_scale


def gemm_fp8(a_q, a_scale, w_q, w_scale, bias=None, out=None, act: int = ACT_NONE, n_split: int = 0, out1=None,
             act1: int = ACT_NONE, gate=None, res=None):
    """out = (a_q @ w_q.T) * a_scale[:, None] * w_scale + bias with hv_gemm_bf16's epilogues; a_q [M,K], w_q [N,K] e4m3fn."""
    for t, nm in ((a_q, "a_q"), (w_q, "w_q")):
        if not t.is_cuda or t.dtype != FP8:
            raise _lib.HVKernelError(f"gemm_fp8: {nm} must be a float8_e4m3fn GPU tensor")
    _chk(a_scale, torch.float32, "a_scale"), _chk(w_scale, BF16, "w_scale")
    m, k, lda = _rows(a_q, "a_q")
    n, kw, ldw = _rows(w_q, "w_q")
    assert k == kw and a_scale.numel() >= m and w_scale.numel() == 1
    n0 = n_split if 0 < n_split < n else n
    if out is None:
        out = torch.empty(m, n0, dtype=BF16, device=a_q.device)
    _chk(out, BF16, "out")
    mo, no, ld0 = _rows(out, "out")
    assert mo == m and no >= n0
    ld1 = 0
    if n0 < n:
        _chk(out1, BF16, "out1")
        m1, n1, ld1 = _rows(out1, "out1")
        assert m1 == m and n1 >= n - n0
    for t, nm in ((bias, "bias"), (gate, "gate")):
        if t is not None:
            _chk(t, BF16, nm)
            assert t.numel() == n and t.is_contiguous()
    ld_res = 0
    if res is not None:
        _chk(res, BF16, "res")
        mr, nr, ld_res = _rows(res, "res")
        assert mr == m and nr == n
    _lib.call("gemm_fp8", a_q, lda, a_scale, w_q, ldw, w_scale, bias, m, n, k, out, ld0, act, n0, out1, ld1, act1, gate, res, ld_res)
    return out


def copy3d(src, dst, n_batch: int, rows: int, cols: int, src_bs: int, src_ld: int, dst_bs: int, dst_ld: int):
    """dst[b][r][:cols] = src[b][r][:cols] with explicit element strides; src/dst are any bf16 GPU tensors whose
    data_ptr() is element (0,0,0) of the region."""
    _chk(src, BF16, "src", False), _chk(dst, BF16, "dst", False)
    _lib.call("copy3d_bf16", src, dst, n_batch, rows, cols, src_bs, src_ld, dst_bs, dst_ld)
    return dst
