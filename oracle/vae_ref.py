"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's 3D causal VAE DECODE path
(hyvideo/vae) in plain PyTorch fp32: CausalConv3d, ResnetBlockCausal3D, UpsampleCausal3D, UNetMidBlockCausal3D (with
the frame-causal attention), UpDecoderBlockCausal3D, DecoderCausal3D and the spatial/temporal tiled decode with
linear blending of AutoencoderKLCausal3D.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Pinned by
tests/golden/vae_*.npz (tools/make_golden_vae.py: the reference's unet_causal_3d_blocks.py / vae.py loaded by file
path and executed on CPU; its tiling methods executed bound to a stand-in object).

PARITY UNPINNED for the mid-block attention (K18): its arithmetic lives in diffusers==0.31.0
(`diffusers.models.attention_processor.Attention`, requirements.txt:2), which is absent here and has no reference
test.  `mid_attention` restates the deprecated-attn-block path from the constructor arguments at
unet_causal_3d_blocks.py:580-592 (GroupNorm(32) over channels -> to_q/to_k/to_v Linear(bias) -> single-head SDPA with
the additive mask of prepare_causal_attention_mask :38-46 -> to_out[0] Linear -> + residual -> / rescale(=1)).

State dicts use the reference key names of DecoderCausal3D under the `decoder.` prefix plus `post_quant_conv.*`
(autoencoder_kl_causal_3d.py:98-115).  `Prec(True)` ("fp16-emulated") rounds to fp16 where production holds fp16
(vae_precision fp16 + autocast fp16, config.py:67-73, pipeline_hunyuan_video.py:1071-1073): conv/linear inputs and
outputs, residual sums; GroupNorm + SiLU run in fp32 (group_norm is an autocast-fp32 op).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class Prec:
    def __init__(self, emulate_fp16: bool = False):
        self.emu = emulate_fp16

    def r(self, x: Tensor) -> Tensor:
        return x.to(torch.float16).to(torch.float32) if self.emu else x


FP32 = Prec(False)


def causal_conv3d(x: Tensor, w: Tensor, b: Optional[Tensor], p: Prec) -> Tensor:
    """CausalConv3d.forward, vae/unet_causal_3d_blocks.py:61-75: replicate pad (W k//2,k//2; H k//2,k//2; T k-1,0)
    then Conv3d stride 1."""
    k = w.shape[-1]
    xp = F.pad(p.r(x), (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return p.r(F.conv3d(xp, p.r(w), None if b is None else p.r(b)))


def group_norm_silu(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-6, silu: bool = True) -> Tensor:
    """nn.GroupNorm(32, C, eps=1e-6, affine) [+ SiLU]: unet_causal_3d_blocks.py:302,323,361-363,399-405; fp32."""
    y = F.group_norm(x.float(), groups, w.float(), b.float(), eps)
    return F.silu(y) if silu else y


def upsample_causal(x: Tensor, factor: Sequence[int]) -> Tensor:
    """UpsampleCausal3D.forward interpolate branch, unet_causal_3d_blocks.py:154-172: first frame nearest x(fh,fw)
    only, remaining frames nearest x(ft,fh,fw); T -> 1 + ft*(T-1)."""
    ft, fh, fw = factor
    first, other = x[:, :, :1], x[:, :, 1:]
    first = first.repeat_interleave(fh, 3).repeat_interleave(fw, 4)
    if other.shape[2] > 0:
        other = other.repeat_interleave(ft, 2).repeat_interleave(fh, 3).repeat_interleave(fw, 4)
        return torch.cat([first, other], 2)
    return first


def resnet_block(sd, pre: str, x: Tensor, p: Prec) -> Tensor:
    """ResnetBlockCausal3D.forward with temb=None, time_embedding_norm="default", output_scale_factor 1
    (unet_causal_3d_blocks.py:349-417)."""
    h = group_norm_silu(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    h = causal_conv3d(h, sd[pre + "conv1.conv.weight"], sd[pre + "conv1.conv.bias"], p)
    h = group_norm_silu(h, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    h = causal_conv3d(h, sd[pre + "conv2.conv.weight"], sd[pre + "conv2.conv.bias"], p)
    if pre + "conv_shortcut.conv.weight" in sd:
        x = causal_conv3d(x, sd[pre + "conv_shortcut.conv.weight"], sd[pre + "conv_shortcut.conv.bias"], p)
    return p.r(p.r(x) + h)


def causal_frame_mask(n_frame: int, n_hw: int) -> Tensor:
    """prepare_causal_attention_mask, unet_causal_3d_blocks.py:38-46 as a boolean "may attend" matrix: token i
    (frame i // n_hw) sees tokens of frames <= its own."""
    f = torch.arange(n_frame * n_hw) // n_hw
    return f[None, :] <= f[:, None]


def mid_attention(sd, pre: str, x: Tensor, p: Prec) -> Tensor:
    """UNetMidBlockCausal3D attention step, unet_causal_3d_blocks.py:654-662 + diffusers Attention (see header:
    PARITY UNPINNED).  x: [B,C,T,H,W]."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, C)
    n = F.group_norm(tok.float().transpose(1, 2), 32, sd[pre + "group_norm.weight"].float(),
                     sd[pre + "group_norm.bias"].float(), 1e-6).transpose(1, 2)
    lin = lambda t, name: p.r(F.linear(p.r(t), p.r(sd[pre + name + ".weight"]), p.r(sd[pre + name + ".bias"])))
    q, k, v = lin(n, "to_q"), lin(n, "to_k"), lin(n, "to_v")
    s = torch.matmul(q, k.transpose(1, 2)) * (1.0 / math.sqrt(C))
    s = s.masked_fill(~causal_frame_mask(T, H * W)[None], float("-inf"))
    a = p.r(torch.matmul(p.r(torch.softmax(s, dim=-1)), v))
    o = lin(a, "to_out.0")
    o = p.r(o + p.r(tok))
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def decoder_config(block_out_channels: Sequence[int], layers_per_block: int = 2, time_compression_ratio: int = 4,
                   spatial_compression_ratio: int = 8):
    """Per up block: (in_ch, out_ch, n_resnets, upsample factor or None) - vae/vae.py:181-217."""
    rev = list(reversed(block_out_channels))
    n = len(block_out_channels)
    ns, nt = int(math.log2(spatial_compression_ratio)), int(math.log2(time_compression_ratio))
    out, prev = [], rev[0]
    for i in range(n):
        oc = rev[i]
        final = i == n - 1
        sp = i < ns
        tm = (i >= n - 1 - nt) and not final
        fac = ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None
        out.append((prev, oc, layers_per_block + 1, fac))
        prev = oc
    return out


def decoder_forward(sd: Dict[str, Tensor], z: Tensor, block_out_channels: Sequence[int], p: Prec = FP32,
                    layers_per_block: int = 2, pre: str = "decoder.") -> Tensor:
    """DecoderCausal3D.forward (vae/vae.py:230-294), norm_type "group", mid_block_add_attention True."""
    h = causal_conv3d(z, sd[pre + "conv_in.conv.weight"], sd[pre + "conv_in.conv.bias"], p)
    h = resnet_block(sd, pre + "mid_block.resnets.0.", h, p)
    h = mid_attention(sd, pre + "mid_block.attentions.0.", h, p)
    h = resnet_block(sd, pre + "mid_block.resnets.1.", h, p)
    for i, (_, _, n_res, fac) in enumerate(decoder_config(block_out_channels, layers_per_block)):
        for j in range(n_res):
            h = resnet_block(sd, f"{pre}up_blocks.{i}.resnets.{j}.", h, p)
        if fac is not None:
            h = upsample_causal(h, fac)
            h = causal_conv3d(h, sd[f"{pre}up_blocks.{i}.upsamplers.0.conv.conv.weight"],
                              sd[f"{pre}up_blocks.{i}.upsamplers.0.conv.conv.bias"], p)
    h = group_norm_silu(h, sd[pre + "conv_norm_out.weight"], sd[pre + "conv_norm_out.bias"])
    return causal_conv3d(h, sd[pre + "conv_out.conv.weight"], sd[pre + "conv_out.conv.bias"], p)


def decode_tile(sd, z: Tensor, block_out_channels, p: Prec) -> Tensor:
    """post_quant_conv (1x1x1) then decoder (autoencoder_kl_causal_3d.py:307-308,447-448,524-525)."""
    z = p.r(F.conv3d(p.r(z), p.r(sd["post_quant_conv.weight"]), p.r(sd["post_quant_conv.bias"])))
    return decoder_forward(sd, z, block_out_channels, p)


def _blend(a: Tensor, b: Tensor, extent: int, dim: int, p: Prec) -> Tensor:
    """blend_v / blend_h / blend_t (autoencoder_kl_causal_3d.py:344-360): writes into b, in place, row by row."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    for y in range(extent):
        ia = [slice(None)] * 5
        ib = [slice(None)] * 5
        ia[dim] = -extent + y
        ib[dim] = y
        b[tuple(ib)] = p.r(p.r(a[tuple(ia)] * (1 - y / extent)) + p.r(b[tuple(ib)] * (y / extent)))
    return b


class TileParams:
    """autoencoder_kl_causal_3d.py:117-132 for the shipped 884-16c-hy VAE (sample_size 256, sample_tsize 64); tests use
    reduced sizes."""

    def __init__(self, sample_size=256, sample_tsize=64, n_blocks=4, time_compression_ratio=4, overlap=0.25):
        self.tile_sample_min_tsize = sample_tsize
        self.tile_latent_min_tsize = sample_tsize // time_compression_ratio
        self.tile_sample_min_size = sample_size
        self.tile_latent_min_size = int(sample_size / (2 ** (n_blocks - 1)))
        self.tile_overlap_factor = overlap


def spatial_tiled_decode(sd, z: Tensor, boc, tp: TileParams, p: Prec) -> Tensor:
    """autoencoder_kl_causal_3d.py:422-469."""
    ov = int(tp.tile_latent_min_size * (1 - tp.tile_overlap_factor))
    ext = int(tp.tile_sample_min_size * tp.tile_overlap_factor)
    lim = tp.tile_sample_min_size - ext
    rows = []
    for i in range(0, z.shape[-2], ov):
        rows.append([decode_tile(sd, z[..., i:i + tp.tile_latent_min_size, j:j + tp.tile_latent_min_size], boc, p)
                     for j in range(0, z.shape[-1], ov)])
    out_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend(rows[i - 1][j], tile, ext, 3, p)
            if j > 0:
                tile = _blend(row[j - 1], tile, ext, 4, p)
            res.append(tile[..., :lim, :lim])
        out_rows.append(torch.cat(res, dim=-1))
    return torch.cat(out_rows, dim=-2)


def temporal_tiled_decode(sd, z: Tensor, boc, tp: TileParams, p: Prec, spatial: bool = True) -> Tensor:
    """autoencoder_kl_causal_3d.py:510-541."""
    T = z.shape[2]
    ov = int(tp.tile_latent_min_tsize * (1 - tp.tile_overlap_factor))
    ext = int(tp.tile_sample_min_tsize * tp.tile_overlap_factor)
    lim = tp.tile_sample_min_tsize - ext
    row = []
    for i in range(0, T, ov):
        tile = z[:, :, i:i + tp.tile_latent_min_tsize + 1]
        if spatial and (tile.shape[-1] > tp.tile_latent_min_size or tile.shape[-2] > tp.tile_latent_min_size):
            dec = spatial_tiled_decode(sd, tile, boc, tp, p)
        else:
            dec = decode_tile(sd, tile, boc, p)
        if i > 0:
            dec = dec[:, :, 1:]
        row.append(dec)
    res = []
    for i, tile in enumerate(row):
        if i > 0:
            tile = _blend(row[i - 1], tile, ext, 2, p)
            res.append(tile[:, :, :lim])
        else:
            res.append(tile[:, :, :lim + 1])
    return torch.cat(res, dim=2)


def decode(sd, z: Tensor, boc, tp: TileParams, p: Prec = FP32, tiling: bool = True) -> Tensor:
    """AutoencoderKLCausal3D._decode (autoencoder_kl_causal_3d.py:298-313) with enable_tiling()."""
    if tiling and z.shape[2] > tp.tile_latent_min_tsize:
        return temporal_tiled_decode(sd, z, boc, tp, p)
    if tiling and (z.shape[-1] > tp.tile_latent_min_size or z.shape[-2] > tp.tile_latent_min_size):
        return spatial_tiled_decode(sd, z, boc, tp, p)
    return decode_tile(sd, z, boc, p)


def postprocess(image: Tensor, p: Prec) -> Tensor:
    """pipeline_hunyuan_video.py:1090-1092: (image / 2 + 0.5).clamp(0, 1) in the VAE dtype, then .float()."""
    return p.r(p.r(image / 2) + 0.5).clamp(0, 1).float()
