"""Tensor-level wrappers over the VAE-decode entry points of the C ABI (include/hv_kernels.h).  fp16, channels-last."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from .ops import _chk

F16 = torch.float16


def _i64x4(v: Sequence[int]):
    return [int(x) for x in v]


def _i32x4(v: Sequence[int]):
    return [int(x) for x in v]


def gemm_f16(a, w, bias=None, out=None, out_f32: bool = False, res=None, n: Optional[int] = None, k: Optional[int] = None):
    """out[M, n] = a[M, k] @ w[n, k]^T + bias (+ res).  a/w/out are 2-D views (row strides free)."""
    _chk(a, F16, "a"), _chk(w, F16, "w")
    m = a.shape[0]
    k = a.shape[1] if k is None else k
    n = w.shape[0] if n is None else n
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32 if out_f32 else F16, device=a.device)
    _chk(out, torch.float32 if out_f32 else F16, "out")
    if bias is not None:
        _chk(bias, F16, "bias")
    ld_res = 0
    if res is not None:
        _chk(res, F16, "res")
        ld_res = res.stride(0)
    _lib.call("gemm_f16", a, a.stride(0), w, w.stride(0), bias, m, n, k, out, out.stride(0),
                                       1 if out_f32 else 0, res, ld_res)
    return out


def _conv_source_limit(x):
    """The conv gather addresses its source with 32-bit byte offsets: one activation must stay below 4 GiB.  Every tile of the
    reference's tiled decode/encode is far below that (largest: 1.09 GB); an untiled call on a large video is what can exceed it."""
    if x.shape[0] * x.stride(0) * 2 >= 2 ** 32:
        raise ValueError(f"activation of {x.shape[0] * x.stride(0) * 2 / 2**30:.1f} GiB exceeds the conv kernel's 4 GiB source limit: "
                         "call vae.enable_tiling() (the reference's default for decode) or decode/encode a smaller clip")


class GNStats:
    """GroupNorm statistics of an activation taken by the conv epilogue that stored it: `partial` fp32 [rows, C, 2]."""
    __slots__ = ("partial", "rows", "M", "C")

    def __init__(self, partial, rows, M, C):
        self.partial, self.rows, self.M, self.C = partial, rows, M, C


GN_FOLD_WS_FLOATS = 16384          # include/hv_kernels.h HV_GN_FOLD_WS_FLOATS: workspace hv_groupnorm_finalize_f16 expects behind the partials


def _gn_stats_buffer(rows: int, c: int, device):
    return torch.empty(rows * c * 2 + GN_FOLD_WS_FLOATS + 2, dtype=torch.float32, device=device)


def conv3d_causal(x, w_taps, bias, T: int, H: int, W: int, cin: int, cout: int, up_t: bool = False, up_hw: bool = False,
                  res=None, out=None, gn_stats: bool = False):
    """x: channels-last source rows [sT*sH*sW, >=cin]; returns [T*H*W, cout] fp16 (with gn_stats: (out, GNStats))."""
    _chk(x, F16, "x"), _chk(w_taps, F16, "w_taps")
    _conv_source_limit(x)
    assert w_taps.is_contiguous() and w_taps.numel() == cout * 27 * cin, (w_taps.shape, cout, cin)
    if out is None:
        out = torch.empty(T * H * W, cout, dtype=F16, device=x.device)
    ld_res = 0
    if res is not None:
        _chk(res, F16, "res")
        ld_res = res.stride(0)
    st = None
    if gn_stats:
        rows = _lib.host("gn_partial_rows", T * H * W)
        st = GNStats(_gn_stats_buffer(rows, cout, x.device), rows, T * H * W, cout)
    _lib.call("conv3d_causal_f16", x, x.stride(0), w_taps, bias, out, out.stride(0), T, H, W,
              cin, cout, int(up_t), int(up_hw), res, ld_res, st.partial if st else None, st.partial.numel() if st else 0)
    return (out, st) if gn_stats else out


def conv_out_mode() -> str:
    """HV_VAE_CONV_OUT = planes (default) | gemm: the decoder's conv_norm_out + SiLU + conv_out as hv_conv3d_cout4_f16 (one streaming
    pass + a gather-sum) or as GroupNorm apply + the narrow implicit-GEMM conv (the form before; kept for A/B)."""
    import os
    m = os.environ.get("HV_VAE_CONV_OUT", "planes")
    if m not in ("planes", "gemm"):
        raise ValueError(f"HV_VAE_CONV_OUT={m!r}: expected planes or gemm")
    return m


def cout4_weight_fragments(w):
    """Conv3d weight [Cout <= 3, Cin <= 128, 3, 3, 3] -> the MFMA fragment order hv_conv3d_cout4_f16 reads, fp16 [7, 4, 64, 8]:
    element (nb, ks, lane, j) = w[c][ks*32 + 8*(lane >> 4) + j][tap] with 4*tap + c = 16*nb + (lane & 15); zero rows for c == 3,
    tap >= 27 and channels >= Cin (include/hv_kernels.h)."""
    co, ci = w.shape[:2]
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and co <= 3 and ci <= 128 and ci % 32 == 0, tuple(w.shape)
    wn = torch.zeros(28, 4, 128, dtype=F16, device=w.device)                # [tap][c][channel]
    wn[:27, :co, :ci] = w.detach().to(F16).reshape(co, ci, 27).permute(2, 0, 1)
    wn = wn.reshape(7, 16, 4, 4, 8)                                         # [nb][row i][ks][lane >> 4][j]
    return wn.permute(0, 2, 3, 1, 4).reshape(7, 4, 64, 8).contiguous()      # lane = (lane >> 4) * 16 + i


def conv_cout4(x, affine, silu: bool, w_frag, bias, T: int, H: int, W: int, cin: int, cout: int, out=None):
    """GroupNorm affine (+ SiLU) + 3x3x3 causal conv to cout <= 3 channels (hv_conv3d_cout4_f16); returns fp16 [T*H*W, 8], columns
    cout.. zero.  affine: fp32 [cin, 2] or None (x is convolved as it is)."""
    _chk(x, F16, "x"), _chk(w_frag, F16, "w_frag"), _chk(bias, F16, "bias")
    if affine is not None:
        _chk(affine, torch.float32, "affine")
    m = T * H * W
    assert x.shape[0] == m and x.shape[1] >= cin and tuple(w_frag.shape) == (7, 4, 64, 8), (x.shape, m, cin, w_frag.shape)
    if out is None:
        out = torch.empty(m, 8, dtype=F16, device=x.device)
    n = _lib.host("conv3d_cout4_planes_floats", m)
    planes = torch.empty(n, dtype=torch.float32, device=x.device)
    _lib.call("conv3d_cout4_f16", x, x.stride(0), affine, int(silu), w_frag, bias, out, out.stride(0), T, H, W, cin, cout, planes, n)
    return out


def subpixel_mode() -> str:
    """HV_VAE_SUBPIXEL = fast (default) | exact | off: how UpsampleCausal3D's conv runs, see subpixel_weights()."""
    import os
    m = os.environ.get("HV_VAE_SUBPIXEL", "fast")
    if m not in ("fast", "exact", "off"):
        raise ValueError(f"HV_VAE_SUBPIXEL={m!r}: expected fast, exact or off")
    return m


def subpixel_weights(w, up_t: bool, mode: str = "fast", cin_pad: Optional[int] = None, cout_pad: Optional[int] = None):
    """Weights + tap table of hv_conv3d_upsampled_subpixel_f16 from a Conv3d weight w [Cout, Cin, 3, 3, 3].

    Nearest x2 upsampling followed by a 3-tap conv along one axis reads, for an output of parity p, only two source samples:
        H/W: p = 0:  w0 x[k-1] + (w1+w2) x[k]         p = 1: (w0+w1) x[k] + w2 x[k+1]
        T  : p = 0:  w0 x[k-1] + (w1+w2) x[k]         p = 1: (w0+w1) x[k-1] + w2 x[k]   (k = kt + 1: frame 2k-1; first frame kept single)
    (unet_causal_3d_blocks.py:154-172 + :49-75).  Per parity class the 27 taps collapse to 2x2x2 (3x2x2 if T is not upsampled).
    The sums are formed in fp64 from the fp16 weights; `fast` rounds each to fp16 (the only difference from the 27-tap form: one
    extra rounding of up to seven of eight weights, <= 2 fp16 ulp at the output), `exact` adds the rounding residue of every summed
    weight as a second tap at the same offset (fp16 products are exact in the fp32 accumulator, so hi + lo reproduces the sum to
    2^-22).  Returns (w_sub [classes, cout_pad, ntap*cin_pad] f16, table int32 [classes, ntap], ntap)."""
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and mode in ("fast", "exact")
    co, ci = w.shape[:2]
    cip, cop = cin_pad or ci, cout_pad or co
    w64 = w.detach().to(F16).to(torch.float64)
    hw = {0: [(-1, (0,)), (0, (1, 2))], 1: [(0, (0, 1)), (1, (2,))]}
    tt = {0: [(-1, (0,)), (0, (1, 2))], 1: [(-1, (0, 1)), (0, (2,))]} if up_t else {0: [(-2, (0,)), (-1, (1,)), (0, (2,))]}
    classes = [(pt, ph, pw) for pt in sorted(tt) for ph in (0, 1) for pw in (0, 1)]
    w_cls, t_cls = [], []
    for pt, ph, pw in classes:
        cols, offs = [], []
        for ot, it in tt[pt]:
            for oh, ih in hw[ph]:
                for ow, iw in hw[pw]:
                    acc = torch.zeros(co, ci, dtype=torch.float64, device=w.device)
                    for a in it:
                        for b in ih:
                            for c in iw:
                                acc += w64[:, :, a, b, c]
                    hi = acc.to(F16)
                    e = (ot + 8) | ((oh + 8) << 4) | ((ow + 8) << 8)
                    cols.append(hi), offs.append(e)
                    if mode == "exact" and len(it) * len(ih) * len(iw) > 1:
                        cols.append((acc - hi.to(torch.float64)).to(F16)), offs.append(e)
        wc = torch.zeros(cop, len(cols), cip, dtype=F16, device=w.device)
        wc[:co, :, :ci] = torch.stack(cols, 1)
        w_cls.append(wc.reshape(cop, -1)), t_cls.append(offs)
    ntap = len(t_cls[0])
    assert all(len(t) == ntap for t in t_cls)
    table = torch.tensor(t_cls, dtype=torch.int32, device=w.device)
    return torch.stack(w_cls, 0).contiguous(), table, ntap


def conv3d_upsampled_subpixel(x, w_sub, table, ntap: int, bias, sT: int, sH: int, sW: int, cin: int, cout: int, up_t: bool, out=None,
                              gn_stats: bool = False):
    """x: channels-last source rows [sT*sH*sW, >=cin] -> [T2*2sH*2sW, cout] fp16, T2 = 2 sT - 1 if up_t else sT."""
    _chk(x, F16, "x"), _chk(w_sub, F16, "w_sub"), _chk(table, torch.int32, "tap_table")
    _conv_source_limit(x)
    ncls = 8 if up_t else 4
    assert w_sub.is_contiguous() and tuple(w_sub.shape) == (ncls, cout, ntap * cin), (w_sub.shape, ncls, cout, ntap, cin)
    assert table.is_contiguous() and tuple(table.shape) == (ncls, ntap)
    T2 = 2 * sT - 1 if up_t else sT
    if out is None:
        out = torch.empty(T2 * 4 * sH * sW, cout, dtype=F16, device=x.device)
    st = None
    if gn_stats:
        rows = _lib.host("subpixel_gn_partial_rows", sT, sH, sW, int(up_t))
        st = GNStats(_gn_stats_buffer(rows, cout, x.device), rows, T2 * 4 * sH * sW, cout)
    _lib.call("conv3d_upsampled_subpixel_f16", x, x.stride(0), w_sub, table, ntap, bias, out, out.stride(0), sT, sH, sW, cin, cout,
              int(up_t), st.partial if st else None, st.partial.numel() if st else 0)
    return (out, st) if gn_stats else out


def conv3d_causal_strided(x, w_taps, bias, sT: int, sH: int, sW: int, cin: int, cout: int, stride=(1, 1, 1)):
    """DownsampleCausal3D conv: x channels-last [sT*sH*sW, >=cin] -> ([T*H*W, cout] fp16, T, H, W)."""
    _chk(x, F16, "x"), _chk(w_taps, F16, "w_taps")
    _conv_source_limit(x)
    assert w_taps.is_contiguous() and w_taps.numel() == cout * 27 * cin, (w_taps.shape, cout, cin)
    st, sh, sw = (int(v) for v in stride)
    T, H, W = (sT - 1) // st + 1, (sH - 1) // sh + 1, (sW - 1) // sw + 1
    out = torch.empty(T * H * W, cout, dtype=F16, device=x.device)
    _lib.call("conv3d_causal_strided_f16", x, x.stride(0), w_taps, bias, out, out.stride(0),
                                                        sT, sH, sW, cin, cout, st, sh, sw)
    return out, T, H, W


def temporal_avg_pool(x, T: int, HW: int, k: int, s: int):
    """t_ops pool: replicate-pad k-1 frames in front + avg over k frames, stride s.  x [T*HW, C] -> ([T'*HW, C], T')."""
    _chk(x, F16, "x")
    t_out = (T - 1) // s + 1
    out = torch.empty(t_out * HW, x.shape[1], dtype=F16, device=x.device)
    _lib.call("temporal_resample_f16", x, x.stride(0), out, out.stride(0), T, HW, x.shape[1], 0, k, s)
    return out, t_out


def temporal_nearest_up(x, T: int, HW: int, s: int):
    """t_ops interp: every frame repeated s times (F.interpolate nearest along T).  x [T*HW, C] -> ([T*s*HW, C], T*s)."""
    _chk(x, F16, "x")
    out = torch.empty(T * s * HW, x.shape[1], dtype=F16, device=x.device)
    _lib.call("temporal_resample_f16", x, x.stride(0), out, out.stride(0), T, HW, x.shape[1], 1, 1, s)
    return out, T * s


_gn_ws = {}


def groupnorm_affine(x, weight, bias, groups: int = 32, eps: float = 1e-6):
    """per-channel (scale, shift) fp32 [C, 2] of GroupNorm(groups) over all rows of x [M, C]."""
    _chk(x, F16, "x"), _chk(weight, F16, "weight"), _chk(bias, F16, "bias")
    m, c = x.shape
    key = (str(x.device), c, torch.cuda.current_stream(x.device).cuda_stream)     # per stream: tiles are decoded on several at once
    ws = _gn_ws.get(key)
    if ws is None:
        ws = torch.empty(1024 * c * 2, dtype=torch.float32, device=x.device)
        _gn_ws[key] = ws
    aff = torch.empty(c, 2, dtype=torch.float32, device=x.device)
    _lib.call("groupnorm_affine_f16", x, x.stride(0), m, c, groups, eps, weight, bias, ws,
                                                   ws.numel(), aff)
    return aff


def groupnorm_affine_from_stats(st: GNStats, weight, bias, groups: int = 32, eps: float = 1e-6):
    """The same affine from statistics a conv epilogue took (conv3d_causal(..., gn_stats=True)): no pass over the activation."""
    _chk(weight, F16, "weight"), _chk(bias, F16, "bias")
    aff = torch.empty(st.C, 2, dtype=torch.float32, device=st.partial.device)
    _lib.call("groupnorm_finalize_f16", st.partial, st.partial.numel(), st.rows, st.M, st.C, groups, eps, weight, bias, aff)
    return aff


def groupnorm_apply(x, affine, silu: bool, out=None):
    _chk(x, F16, "x"), _chk(affine, torch.float32, "affine")
    m, c = x.shape
    if out is None:
        out = torch.empty(m, c, dtype=F16, device=x.device)
    _lib.call("groupnorm_apply_f16", x, x.stride(0), out, out.stride(0), m, c, affine, int(silu))
    return out


def softmax_rows(s_f32, cols: int, cols_pad: int, scale: float, out=None, causal_block: int = 0):
    """P = softmax(scale * S) per row over the first `cols` columns (causal_block = HW: over the keys of frames <= the row's frame)."""
    _chk(s_f32, torch.float32, "S")
    rows = s_f32.shape[0]
    if out is None:
        out = torch.empty(rows, cols_pad, dtype=F16, device=s_f32.device)
    _lib.call("softmax_rows_f32_f16", s_f32, s_f32.stride(0), out, out.stride(0), rows, cols, cols_pad, scale, causal_block)
    return out


def transpose_16b(src, dst):
    """dst[c, r] = src[r, c] for 2-D 16-bit views."""
    assert src.element_size() == 2 and dst.element_size() == 2 and src.is_cuda and dst.is_cuda
    r, c = src.shape
    _lib.call("transpose_16b", src, src.stride(0), dst, dst.stride(0), r, c)
    return dst


def latent_tile(z_f32_view, cpad: int):
    """z view [C,T,H,W] fp32 (any strides) -> channels-last fp16 [T*H*W, cpad]."""
    _chk(z_f32_view, torch.float32, "z", False)
    c, t, h, w = z_f32_view.shape
    sc, st, sh, sw = z_f32_view.stride()
    out = torch.empty(t * h * w, cpad, dtype=F16, device=z_f32_view.device)
    _lib.call("vae_latent_tile_f16", z_f32_view, sc, st, sh, sw, c, t, h, w, cpad, out)
    return out


def blend_(a_view, b_view, axis: int, extent: int):
    """b = a*(1-y/extent) + b*(y/extent) along `axis` of equal-shaped strided 4-D fp16 views [C,T,H,W]."""
    _chk(a_view, F16, "a", False), _chk(b_view, F16, "b", False)
    assert a_view.shape == b_view.shape and a_view.dim() == 4
    if b_view.numel() == 0:
        return b_view          # empty overlap (e.g. a trailing temporal tile of a single latent frame): nothing to blend
    _lib.call("vae_blend_f16", a_view, _i64x4(a_view.stride()), b_view, _i64x4(b_view.stride()),
                                            _i32x4(b_view.shape), axis, extent)
    return b_view


def copy4d_(src_view, dst_view):
    assert src_view.shape == dst_view.shape and src_view.dim() == 4 and src_view.element_size() == 2 and dst_view.element_size() == 2
    assert src_view.is_cuda and dst_view.is_cuda
    if dst_view.numel() == 0:
        return dst_view
    _lib.call("copy4d_16b", src_view, _i64x4(src_view.stride()), dst_view, _i64x4(dst_view.stride()),
                                         _i32x4(dst_view.shape))
    return dst_view


def postprocess(x_f16):
    _chk(x_f16, F16, "x")
    assert x_f16.is_contiguous()
    out = torch.empty(x_f16.shape, dtype=torch.float32, device=x_f16.device)
    _lib.call("vae_postprocess_f16_f32", x_f16, out, x_f16.numel())
    return out
