"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's VAE ENCODE path and of the fork's
temporal-ops ("t_ops") hooks on both halves of the VAE, in plain PyTorch fp32:

  EncoderCausal3D (hyvideo/vae/vae.py:32-136), DownEncoderBlockCausal3D / DownsampleCausal3D
  (vae/unet_causal_3d_blocks.py:680-790,185-247), quant_conv + DiagonalGaussianDistribution (vae/vae.py:297-358,
  autoencoder_kl_causal_3d.py:110-115,259-296), spatial/temporal tiled ENCODE (autoencoder_kl_causal_3d.py:362-420,
  470-510), AutoencoderKLCausal3D.forward (:545-580), and the t_ops hooks: temporal avg-pool before/after each encoder /
  mid-block resnet, `downsample_stride` override, temporal nearest interpolation before/after each decoder resnet
  (unet_causal_3d_blocks.py:622-672,736-790,853-912; vae/__init__.py:15-63; t_ops_config.json).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Pinned by
tests/golden/vae_enc_*.npz (tools/make_golden_vae_enc.py: the reference's own modules executed on CPU).  The mid-block
attention is the same restatement as in vae_ref.py (PARITY UNPINNED there: diffusers is absent); the encoder goldens include
it, so encoder parity through the mid block is pinned only up to that restatement being shared by both sides of the fixture
generator (see tools/make_golden_vae_enc.py)."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .vae_ref import (FP32, Prec, TileParams, _blend, causal_conv3d, decoder_config, group_norm_silu, mid_attention,
                      resnet_block, upsample_causal)

Tensor = torch.Tensor


def causal_conv3d_strided(x: Tensor, w: Tensor, b: Optional[Tensor], stride: Sequence[int], p: Prec) -> Tensor:
    """CausalConv3d with stride (unet_causal_3d_blocks.py:61-75): the same replicate padding, then Conv3d(stride)."""
    k = w.shape[-1]
    xp = F.pad(p.r(x), (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return p.r(F.conv3d(xp, p.r(w), None if b is None else p.r(b), stride=tuple(stride)))


def t_pool(x: Tensor, k: int, s: int, p: Prec) -> Tensor:
    """The fork's temporal pool (unet_causal_3d_blocks.py:659-662,768-770): replicate-pad k-1 frames in front, avg_pool3d."""
    xp = F.pad(x, (0, 0, 0, 0, k - 1, 0), mode="replicate")
    return p.r(F.avg_pool3d(xp, kernel_size=(k, 1, 1), stride=(s, 1, 1)))


def t_interp(x: Tensor, sc: int, mode: str = "nearest") -> Tensor:
    """unet_causal_3d_blocks.py:889-895: F.interpolate(scale_factor=(sc,1,1), mode)."""
    if x.shape[2] == 0:
        return x
    return F.interpolate(x, scale_factor=(sc, 1, 1), mode=mode)


def encoder_config(block_out_channels: Sequence[int], time_compression_ratio: int = 4, spatial_compression_ratio: int = 8):
    """Per down block: (in_ch, out_ch, downsample stride or None) - vae/vae.py:63-100."""
    n = len(block_out_channels)
    ns, nt = int(math.log2(spatial_compression_ratio)), int(math.log2(time_compression_ratio))
    out, prev = [], block_out_channels[0]
    for i in range(n):
        oc = block_out_channels[i]
        final = i == n - 1
        sp = i < ns
        tm = (i >= n - 1 - nt) and not final
        out.append((prev, oc, ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1)) if (sp or tm) else None))
        prev = oc
    return out


def _pool_conf(block_cfg: Optional[dict], n_res: int, default_k: int = 2):
    """apply_t_ops_config / apply_t_ops_config_midblock: per-resnet (before, after, k, s)."""
    if not block_cfg:
        return [(False, False, 0, 0)] * n_res
    epb, epa = block_cfg.get("enable_t_pool_before_block", []), block_cfg.get("enable_t_pool_after_block", [])
    if len(epb) != n_res or len(epa) != n_res:
        raise ValueError(f"T-ops config mismatch: expecting {n_res} bools in each list")
    k, s = block_cfg.get("pool_t_kernel", default_k), block_cfg.get("pool_t_stride", 2)
    return [(bool(epb[i]), bool(epa[i]), k, s) for i in range(n_res)]


def _by_index(cfgs, i):
    for c in cfgs or []:
        if c.get("block_index") == i:
            return c
    return None


def mid_block(sd, pre: str, h: Tensor, p: Prec, cfg: Optional[dict] = None) -> Tensor:
    """UNetMidBlockCausal3D.forward (unet_causal_3d_blocks.py:647-674): resnet 0, then attention + resnet 1; the fork's
    pools wrap each resnet (the attention of resnet 1 runs BEFORE its `before` pool)."""
    conf = _pool_conf(cfg if (cfg and "enable_t_pool_before_block" in cfg) else None, 2)
    for i in range(2):
        if i > 0:
            h = mid_attention(sd, pre + "attentions.0.", h, p)
        before, after, k, s = conf[i]
        if before:
            h = t_pool(h, k, s, p)
        h = resnet_block(sd, f"{pre}resnets.{i}.", h, p)
        if after:
            h = t_pool(h, k, s, p)
    return h


def encoder_forward(sd: Dict[str, Tensor], x: Tensor, block_out_channels: Sequence[int], p: Prec = FP32, layers_per_block: int = 2,
                    t_ops: Optional[dict] = None, pre: str = "encoder.") -> Tensor:
    """EncoderCausal3D.forward (vae/vae.py:116-136) with optional t_ops["encoder"]."""
    enc = (t_ops or {}).get("encoder", {})
    h = causal_conv3d(x, sd[pre + "conv_in.conv.weight"], sd[pre + "conv_in.conv.bias"], p)
    for i, (_, _, stride) in enumerate(encoder_config(block_out_channels)):
        bc = _by_index(enc.get("down_blocks"), i)
        conf = _pool_conf(bc, layers_per_block)
        for j in range(layers_per_block):
            before, after, k, s = conf[j]
            if before:
                h = t_pool(h, k, s, p)
            h = resnet_block(sd, f"{pre}down_blocks.{i}.resnets.{j}.", h, p)
            if after:
                h = t_pool(h, k, s, p)
        if stride is not None:
            if bc and "downsample_stride" in bc:
                stride = tuple(bc["downsample_stride"])
            h = causal_conv3d_strided(h, sd[f"{pre}down_blocks.{i}.downsamplers.0.conv.conv.weight"],
                                      sd[f"{pre}down_blocks.{i}.downsamplers.0.conv.conv.bias"], stride, p)
    h = mid_block(sd, pre + "mid_block.", h, p, enc.get("mid_block"))
    h = group_norm_silu(h, sd[pre + "conv_norm_out.weight"], sd[pre + "conv_norm_out.bias"])
    return causal_conv3d(h, sd[pre + "conv_out.conv.weight"], sd[pre + "conv_out.conv.bias"], p)


def encode_tile(sd, x: Tensor, boc, p: Prec, t_ops: Optional[dict] = None) -> Tensor:
    """encoder + quant_conv (1x1x1) -> moments [B, 2*latent, T', H', W'] (autoencoder_kl_causal_3d.py:289-292,398-399)."""
    h = encoder_forward(sd, x, boc, p, t_ops=t_ops)
    return p.r(F.conv3d(p.r(h), p.r(sd["quant_conv.weight"]), p.r(sd["quant_conv.bias"])))


def decoder_forward_tops(sd: Dict[str, Tensor], z: Tensor, block_out_channels: Sequence[int], p: Prec = FP32,
                         layers_per_block: int = 2, t_ops: Optional[dict] = None, pre: str = "decoder.") -> Tensor:
    """DecoderCausal3D.forward with t_ops["decoder"]: nearest temporal interpolation before/after each up-block resnet
    (unet_causal_3d_blocks.py:876-912) and pools around the mid-block resnets."""
    dec = (t_ops or {}).get("decoder", {})
    h = causal_conv3d(z, sd[pre + "conv_in.conv.weight"], sd[pre + "conv_in.conv.bias"], p)
    h = mid_block(sd, pre + "mid_block.", h, p, dec.get("mid_block"))
    for i, (_, _, n_res, fac) in enumerate(decoder_config(block_out_channels, layers_per_block)):
        bc = _by_index(dec.get("up_blocks"), i)
        eib = (bc or {}).get("enable_t_interp_before_block", [False] * n_res)
        eia = (bc or {}).get("enable_t_interp_after_block", [False] * n_res)
        if len(eib) != n_res or len(eia) != n_res:
            raise ValueError(f"[UpDecoderBlockCausal3D] config mismatch: expecting {n_res} bools in each list.")
        sc, mode = (bc or {}).get("interp_t_scale_factor", 2), (bc or {}).get("interp_mode", "nearest")
        for j in range(n_res):
            if eib[j]:
                h = t_interp(h, sc, mode)
            h = resnet_block(sd, f"{pre}up_blocks.{i}.resnets.{j}.", h, p)
            if eia[j]:
                h = t_interp(h, sc, mode)
        if fac is not None:
            h = upsample_causal(h, fac)
            h = causal_conv3d(h, sd[f"{pre}up_blocks.{i}.upsamplers.0.conv.conv.weight"],
                              sd[f"{pre}up_blocks.{i}.upsamplers.0.conv.conv.bias"], p)
    h = group_norm_silu(h, sd[pre + "conv_norm_out.weight"], sd[pre + "conv_norm_out.bias"])
    return causal_conv3d(h, sd[pre + "conv_out.conv.weight"], sd[pre + "conv_out.conv.bias"], p)


def decode_tile_tops(sd, z: Tensor, boc, p: Prec, t_ops: Optional[dict] = None) -> Tensor:
    z = p.r(F.conv3d(p.r(z), p.r(sd["post_quant_conv.weight"]), p.r(sd["post_quant_conv.bias"])))
    return decoder_forward_tops(sd, z, boc, p, t_ops=t_ops)


# ----------------------------------------------------------------------------- tiled encode
def spatial_tiled_encode(sd, x: Tensor, boc, tp: TileParams, p: Prec) -> Tensor:
    """autoencoder_kl_causal_3d.py:362-420 -> moments."""
    ov = int(tp.tile_sample_min_size * (1 - tp.tile_overlap_factor))
    ext = int(tp.tile_latent_min_size * tp.tile_overlap_factor)
    lim = tp.tile_latent_min_size - ext
    rows = []
    for i in range(0, x.shape[-2], ov):
        rows.append([encode_tile(sd, x[..., i:i + tp.tile_sample_min_size, j:j + tp.tile_sample_min_size], boc, p)
                     for j in range(0, x.shape[-1], ov)])
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


def temporal_tiled_encode(sd, x: Tensor, boc, tp: TileParams, p: Prec, spatial: bool = True) -> Tensor:
    """autoencoder_kl_causal_3d.py:470-510 -> moments."""
    T = x.shape[2]
    ov = int(tp.tile_sample_min_tsize * (1 - tp.tile_overlap_factor))
    ext = int(tp.tile_latent_min_tsize * tp.tile_overlap_factor)
    lim = tp.tile_latent_min_tsize - ext
    row = []
    for i in range(0, T, ov):
        tile = x[:, :, i:i + tp.tile_sample_min_tsize + 1]
        if spatial and (tile.shape[-1] > tp.tile_sample_min_size or tile.shape[-2] > tp.tile_sample_min_size):
            tile = spatial_tiled_encode(sd, tile, boc, tp, p)
        else:
            tile = encode_tile(sd, tile, boc, p)
        if i > 0:
            tile = tile[:, :, 1:]
        row.append(tile)
    res = []
    for i, tile in enumerate(row):
        if i > 0:
            tile = _blend(row[i - 1], tile, ext, 2, p)
            res.append(tile[:, :, :lim])
        else:
            res.append(tile[:, :, :lim + 1])
    return torch.cat(res, dim=2)


def encode(sd, x: Tensor, boc, tp: TileParams, p: Prec = FP32, tiling: bool = False, t_ops: Optional[dict] = None) -> Tensor:
    """AutoencoderKLCausal3D.encode (autoencoder_kl_causal_3d.py:259-296) -> moments (the posterior's parameters)."""
    if tiling and x.shape[2] > tp.tile_sample_min_tsize:
        return temporal_tiled_encode(sd, x, boc, tp, p)
    if tiling and (x.shape[-1] > tp.tile_sample_min_size or x.shape[-2] > tp.tile_sample_min_size):
        return spatial_tiled_encode(sd, x, boc, tp, p)
    return encode_tile(sd, x, boc, p, t_ops)


# ----------------------------------------------------------------------------- posterior
def posterior(moments: Tensor):
    """DiagonalGaussianDistribution (vae/vae.py:297-330): mean, logvar = chunk(2, dim=1); logvar clamped to [-30, 20];
    std = exp(0.5 logvar).  Returns (mean, logvar, std)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean, logvar, torch.exp(0.5 * logvar)


def posterior_sample(moments: Tensor, noise: Tensor) -> Tensor:
    """.sample(): mean + std * randn (vae/vae.py:320-330) with the noise made explicit."""
    mean, _, std = posterior(moments)
    return mean + std * noise


def posterior_kl(moments: Tensor) -> Tensor:
    """.kl() against the standard normal (vae/vae.py:332-346): 0.5 * sum(mean^2 + var - 1 - logvar) over all non-batch dims."""
    mean, logvar, _ = posterior(moments)
    return 0.5 * torch.sum(mean ** 2 + logvar.exp() - 1.0 - logvar, dim=list(range(1, mean.dim())))


def vae_forward(sd, x: Tensor, boc, p: Prec = FP32, t_ops: Optional[dict] = None) -> Tensor:
    """AutoencoderKLCausal3D.forward with sample_posterior=False (autoencoder_kl_causal_3d.py:545-580; infer.py:52-57):
    encode -> posterior.mode() -> decode."""
    z = posterior(encode_tile(sd, x, boc, p, t_ops))[0]
    return decode_tile_tops(sd, p.r(z), boc, p, t_ops)
