"""AutoencoderKLCausal3D - decode side - on the MI355X kernels, with the reference's call surface
(hyvideo/vae/autoencoder_kl_causal_3d.py: decode(z, return_dict, generator)[0], enable_tiling(), .config.*, .dtype)
and the reference's state-dict key names for `decoder.*` and `post_quant_conv.*` (vae/vae.py:139-226,
vae/unet_causal_3d_blocks.py).  The encode half (EncoderCausal3D, quant_conv, the posterior, tiled
encode, forward(), and the fork's temporal-op hooks of t_ops_config.json on both halves) is built with `with_encoder=True`;
without it checkpoint keys under `encoder.` / `quant_conv.` are ignored on load.

Execution model: activations are fp16 CHANNELS-LAST ([T*H*W voxels, C]) so that
  * CausalConv3d is an implicit GEMM whose A rows are contiguous channel vectors (replicate padding = index clamp,
    nearest upsample = index halving, both inside the DMA gather: K15+K17 cost no extra memory pass),
  * GroupNorm is a per-channel affine after a two-level reduction (K16),
  * the mid-block attention needs no rearrange: "b (f h w) c" IS this layout (K18); the frame-causal mask is realised by
    giving frame f's queries exactly the keys of frames <= f (no [L,L] mask tensor - the reference builds a 606 MB one),
  * tiles are decoded one after another and blended in the reference's exact order (K19) by strided kernels.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import synthetic as syn
from .. import vae_ops as V

F16 = torch.float16


dist = None      # torch.distributed, imported on first use (tests/test_lazy_collectives_gloo.py substitutes a lazy-completion proxy)


def _dist():
    global dist
    if dist is None:
        import torch.distributed as _d
        dist = _d
    return dist


def _r(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _build_tree(root: nn.Module, shapes: Dict[str, Tuple[int, ...]], device, dtype):
    """Create nested parameter holders so that root.state_dict() has exactly the keys of `shapes`."""
    for key, shp in shapes.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], nn.Parameter(torch.empty(*shp, device=device, dtype=dtype), requires_grad=False))


class DiagonalGaussianDistribution:
    """vae/vae.py:297-358 with the reference's attribute and method names.  The posterior's parameters are a few hundred KB
    (the latent); its elementwise arithmetic is torch tensor math on the device that holds `parameters`, not a kernel."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        if parameters.ndim == 3:
            dim = 2
        elif parameters.ndim in (4, 5):
            dim = 1
        else:
            raise NotImplementedError
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=dim)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def kl(self, other: "DiagonalGaussianDistribution" = None) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        dims = list(range(1, self.mean.ndim))
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=dims)
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar
                               + other.logvar, dim=dims)

    def nll(self, sample: torch.Tensor, dims=(1, 2, 3)) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        return 0.5 * torch.sum(math.log(2.0 * math.pi) + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=list(dims))

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLCausal3D(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types=("DownEncoderBlockCausal3D",) * 4,
                 up_block_types=("UpDecoderBlockCausal3D",) * 4, block_out_channels=syn.VAE_BLOCK_OUT_CHANNELS,
                 layers_per_block: int = 2, act_fn: str = "silu", latent_channels: int = 16, norm_num_groups: int = 32,
                 sample_size: int = 256, sample_tsize: int = 64, scaling_factor: float = 0.476986,
                 force_upcast: bool = True, spatial_compression_ratio: int = 8, time_compression_ratio: int = 4,
                 mid_block_add_attention: bool = True, device=None, dtype=F16, with_encoder: bool = False):
        """with_encoder: also hold `encoder.*` / `quant_conv.*` (encode(), forward(): SURVEY.md 8f row 3).  Without it the
        module is the decode half the denoise pipeline needs and checkpoint keys of the encode half are ignored on load."""
        super().__init__()
        if act_fn not in ("silu", "swish") or norm_num_groups != 32 or time_compression_ratio != 4 or \
                spatial_compression_ratio != 8 or not mid_block_add_attention or len(block_out_channels) != 4:
            raise NotImplementedError("kernels cover the shipped 884 VAE topology (SiLU, GroupNorm(32), 4 blocks, mid attention)")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, down_block_types=down_block_types,
                                      up_block_types=up_block_types, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                                      norm_num_groups=norm_num_groups, sample_size=sample_size, sample_tsize=sample_tsize,
                                      scaling_factor=scaling_factor, force_upcast=force_upcast,
                                      spatial_compression_ratio=spatial_compression_ratio,
                                      time_compression_ratio=time_compression_ratio, mid_block_add_attention=mid_block_add_attention)
        self.time_compression_ratio = time_compression_ratio
        self._shapes = dict(syn.vae_decoder_param_shapes(block_out_channels, latent_channels, out_channels, layers_per_block))
        self.with_encoder = with_encoder
        if with_encoder:
            self._shapes.update(syn.vae_encoder_param_shapes(block_out_channels, latent_channels, in_channels, layers_per_block))
        _build_tree(self, self._shapes, device, dtype)
        self._t_ops = None
        self.use_slicing = False
        self.use_spatial_tiling = False
        self.use_temporal_tiling = False
        # autoencoder_kl_causal_3d.py:117-132
        self.tile_sample_min_tsize = sample_tsize
        self.tile_latent_min_tsize = sample_tsize // time_compression_ratio
        self.tile_sample_min_size = sample_size
        self.tile_latent_min_size = int(sample_size / (2 ** (len(block_out_channels) - 1)))
        self.tile_overlap_factor = 0.25
        self._prep = None
        # mid-block attention: all frames of a tile in one score matrix while it stays below this size (a 17x32x32 latent tile:
        # 1.2 GB fp32); larger inputs (untiled decode of a long clip) fall back to one frame of query rows at a time
        self.mid_attention_batch_bytes = 4 << 30
        # tiled decode: independent tiles are decoded on this many HIP streams at once (1 = strictly one after the other).  Same box:
        # 3.50 / 3.48 s on one stream, 3.20 / 3.20 s on two, 3.69 s on three (tools/bench_vae_streams.py; bit-identical outputs)
        self.decode_streams = 2
        self._streams = None

    # ------------------------------------------------------------------ reference surface
    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def enable_temporal_tiling(self, use_tiling: bool = True):
        self.use_temporal_tiling = use_tiling

    def disable_temporal_tiling(self):
        self.enable_temporal_tiling(False)

    def enable_spatial_tiling(self, use_tiling: bool = True):
        self.use_spatial_tiling = use_tiling

    def disable_spatial_tiling(self):
        self.enable_spatial_tiling(False)

    def enable_tiling(self, use_tiling: bool = True):
        self.enable_spatial_tiling(use_tiling)
        self.enable_temporal_tiling(use_tiling)

    def disable_tiling(self):
        self.disable_spatial_tiling()
        self.disable_temporal_tiling()

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        kept = state_dict if self.with_encoder else \
            {k: v for k, v in state_dict.items() if not (k.startswith("encoder.") or k.startswith("quant_conv."))}
        self._prep = None
        return super().load_state_dict(kept, strict=strict, assign=assign)

    # ------------------------------------------------------------------ weight preparation (once)
    def _prepare(self):
        if self._prep is not None:
            return self._prep
        sd = {k: p.detach() for k, p in self.named_parameters()}
        dev = self.device
        P = {}

        def conv3(name):
            w, b = sd[name + ".weight"].to(F16), sd[name + ".bias"].to(F16)
            co, ci = w.shape[0], w.shape[1]
            cop, cip = _r(co, 8), _r(ci, 64)
            wt = torch.zeros(cop, 27, cip, dtype=F16, device=dev)
            wt[:co, :, :ci] = w.permute(0, 2, 3, 4, 1).reshape(co, 27, ci)      # [Cout][(dt,dh,dw)][Cin]
            bp = torch.zeros(cop, dtype=F16, device=dev)
            bp[:co] = b
            P[name] = (wt.contiguous(), bp, cip, cop)

        def lin(name, w, b, kpad=64):
            w, b = w.to(F16), b.to(F16)
            n, k = w.shape
            wp = torch.zeros(_r(n, 8), _r(k, kpad), dtype=F16, device=dev)
            wp[:n, :k] = w
            bp = torch.zeros(_r(n, 8), dtype=F16, device=dev)
            bp[:n] = b
            P[name] = (wp, bp)

        for k in sd:
            if k.endswith(".weight") and sd[k].dim() == 5:
                name = k[:-len(".weight")]
                if sd[k].shape[-1] == 3:
                    conv3(name)
                else:
                    lin(name, sd[k].reshape(sd[k].shape[0], -1), sd[name + ".bias"])
            elif k.endswith(".weight") and sd[k].dim() == 1:
                name = k[:-len(".weight")]
                P[name] = (sd[k].to(F16).contiguous(), sd[name + ".bias"].to(F16).contiguous())
        for half in (("decoder", "encoder") if self.with_encoder else ("decoder",)):
            a = half + ".mid_block.attentions.0."
            wqkv = torch.cat([sd[a + n + ".weight"] for n in ("to_q", "to_k", "to_v")], 0)
            bqkv = torch.cat([sd[a + n + ".bias"] for n in ("to_q", "to_k", "to_v")], 0)
            lin(a + "qkv", wqkv, bqkv)
            lin(a + "to_out.0", sd[a + "to_out.0.weight"], sd[a + "to_out.0.bias"])
        # UpsampleCausal3D convs of the decoder in sub-pixel form (vae_ops.subpixel_weights): the 256- and 512-channel upsamplers
        mode = V.subpixel_mode()
        boc = self.config.block_out_channels
        nb = len(boc)
        for i in range(nb):
            name = f"decoder.up_blocks.{i}.upsamplers.0.conv.conv"
            sp, tm = i < 3, (i >= nb - 1 - 2) and (i != nb - 1)
            if mode == "off" or name not in P or not sp:
                continue
            _, bp, cip, cop = P[name]
            if cip >= 256 and (cip & (cip - 1)) == 0 and cop > 128:
                w_sub, table, ntap = V.subpixel_weights(sd[name + ".weight"].to(dev), tm, mode, cip, cop)
                P[name + "#subpixel"] = (w_sub, table, ntap, bp, cip, cop)
        # the decoder's tail (conv_norm_out + SiLU + conv_out) as one streaming pass + a gather-sum (vae_ops.conv_cout4)
        name = "decoder.conv_out.conv"
        if V.conv_out_mode() == "planes" and name in P:
            w = sd[name + ".weight"]
            if w.shape[0] <= 3 and w.shape[1] <= 128 and w.shape[1] % 32 == 0:
                P[name + "#cout4"] = (V.cout4_weight_fragments(w.to(dev)), P[name][1], int(w.shape[1]), int(w.shape[0]))
        self._prep = P
        return P

    # ------------------------------------------------------------------ decoder on one latent tile
    @staticmethod
    def _pad_channels(x, c_pad):
        """Channel counts that are not a multiple of 64 (reduced test models only; the shipped VAE has 128/256/512):
        zero-pad the rows so the K=64 DMA pieces never run into the next voxel."""
        if x.shape[1] >= c_pad:
            return x
        xp = torch.zeros(x.shape[0], c_pad, dtype=F16, device=x.device)
        V.copy4d_(x[None, None], xp[None, None, :, :x.shape[1]])
        return xp

    def _conv(self, P, name, x, T, H, W, up_t=False, up_hw=False, res=None, gn_stats=False):
        """gn_stats: also return the GroupNorm(32) statistics of the output, taken in the conv's epilogue (None where the epilogue's
        pairwise bookkeeping does not apply: an odd number of channels per group - reduced test models only)."""
        wt, b, cip, cop = P[name]
        if gn_stats and (cop % 64):
            return V.conv3d_causal(self._pad_channels(x, cip), wt, b, T, H, W, cip, cop, up_t, up_hw, res), None
        return V.conv3d_causal(self._pad_channels(x, cip), wt, b, T, H, W, cip, cop, up_t, up_hw, res, gn_stats=gn_stats)

    def _gn(self, P, name, x, silu=True, stats=None):
        """GroupNorm(32) [+ SiLU].  `stats`: the statistics of x if the conv that produced it took them in its epilogue
        (vae_ops.GNStats) - then no pass over x is needed before the normalising one."""
        w, b = P[name]
        if stats is not None:
            assert stats.M == x.shape[0] and stats.C == x.shape[1], (stats.M, stats.C, x.shape)
            return V.groupnorm_apply(x, V.groupnorm_affine_from_stats(stats, w, b, 32, 1e-6), silu)
        return V.groupnorm_apply(x, V.groupnorm_affine(x, w, b, 32, 1e-6), silu)

    def _resnet(self, P, pre, x, T, H, W, x_stats=None, out_stats=False):
        """ResnetBlockCausal3D (unet_causal_3d_blocks.py:395-415).  x_stats: GroupNorm statistics of x if its producer took them;
        out_stats: return (y, statistics of y) for the GroupNorm that consumes y next."""
        h = self._gn(P, pre + "norm1", x, stats=x_stats)
        h, st = self._conv(P, pre + "conv1.conv", h, T, H, W, gn_stats=True)
        h = self._gn(P, pre + "norm2", h, stats=st)
        if (pre + "conv_shortcut.conv") in P:
            ws, bs = P[pre + "conv_shortcut.conv"]
            x = V.gemm_f16(self._pad_channels(x, ws.shape[1]), ws, bs, k=ws.shape[1])
        return self._conv(P, pre + "conv2.conv", h, T, H, W, res=x, gn_stats=out_stats)

    def _mid_attention(self, P, pre, x, T, HW):
        L, C = x.shape
        n = self._gn(P, pre + "group_norm", x, silu=False)
        wqkv, bqkv = P[pre + "qkv"]
        qkv = torch.zeros(_r(L, 8) + 8, 3 * C, dtype=F16, device=x.device)
        V.gemm_f16(n, wqkv, bqkv, out=qkv[:L])
        Lp = _r(L, 64)
        vT = torch.zeros(C, Lp, dtype=F16, device=x.device)
        V.transpose_16b(qkv[:L, 2 * C:], vT)
        a = torch.empty(L, C, dtype=F16, device=x.device)
        scale = 1.0 / math.sqrt(C)
        if L * _r(L, 8) * 4 <= self.mid_attention_batch_bytes:
            # all frames at once: one score GEMM [L, L] (fp32), one frame-causal softmax, one P.V GEMM - the blocks above the
            # frame diagonal are computed and masked (0.3 TFLOP per 17x32x32 tile, a fraction of a millisecond) instead of 3 T
            # small launches whose P.V GEMMs occupy 8 workgroups each
            S = torch.empty(L, _r(L, 8), dtype=torch.float32, device=x.device)
            V.gemm_f16(qkv[:L, :C], qkv[:, C:2 * C], None, out=S, out_f32=True, n=_r(L, 8), k=C)
            Pm = V.softmax_rows(S, L, Lp, scale, causal_block=HW)
            del S
            V.gemm_f16(Pm, vT, None, out=a, n=C, k=Lp)
            wo, bo = P[pre + "to_out.0"]
            return V.gemm_f16(a, wo, bo, res=x, k=wo.shape[1])
        S = torch.empty(HW, _r(L, 8), dtype=torch.float32, device=x.device)
        Pm = torch.empty(HW, Lp, dtype=F16, device=x.device)
        for f in range(T):
            nk = (f + 1) * HW
            nk8, nk64 = _r(nk, 8), _r(nk, 64)
            q = qkv[f * HW:(f + 1) * HW, :C]
            V.gemm_f16(q, qkv[:, C:2 * C], None, out=S, out_f32=True, n=nk8, k=C)
            V.softmax_rows(S, nk, nk64, scale, out=Pm)
            V.gemm_f16(Pm, vT, None, out=a[f * HW:(f + 1) * HW], n=C, k=nk64)
        wo, bo = P[pre + "to_out.0"]
        return V.gemm_f16(a, wo, bo, res=x, k=wo.shape[1])

    def _decode_tile(self, z_view: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
        """z_view: fp32 [C,T,H,W] strided view -> (channels-last fp16 [T'*H'*W', 8] (3 valid channels), T', H', W')."""
        P = self._prepare()
        c, T, H, W = z_view.shape
        x = V.latent_tile(z_view, 64)
        wpq, bpq = P["post_quant_conv"]
        x1 = torch.zeros(T * H * W, 64, dtype=F16, device=x.device)
        V.gemm_f16(x, wpq, bpq, out=x1, n=wpq.shape[0], k=64)
        pre = "decoder."
        h = self._conv(P, pre + "conv_in.conv", x1, T, H, W)
        dec_ops = (self._t_ops or {}).get("decoder", {})
        h, T = self._mid_block(P, pre + "mid_block.", h, T, H, W, dec_ops.get("mid_block"))
        boc = self.config.block_out_channels
        nb = len(boc)
        n_res = self.config.layers_per_block + 1
        st = None          # GroupNorm statistics of h, when the conv that produced h took them
        for i in range(nb):
            bc = self._block_cfg(dec_ops.get("up_blocks"), i) or {}
            eib = bc.get("enable_t_interp_before_block", [False] * n_res)
            eia = bc.get("enable_t_interp_after_block", [False] * n_res)
            if len(eib) != n_res or len(eia) != n_res:
                raise ValueError(f"[UpDecoderBlockCausal3D] config mismatch: expecting {n_res} bools in each list.")
            sc = int(bc.get("interp_t_scale_factor", 2))
            if (any(eib) or any(eia)) and bc.get("interp_mode", "nearest") != "nearest":
                raise NotImplementedError("t_ops interp_mode: only 'nearest' has a kernel (the fork's config default)")
            sp = i < 3
            tm = (i >= nb - 1 - 2) and (i != nb - 1)
            for j in range(n_res):
                if eib[j]:
                    h, T = V.temporal_nearest_up(h, T, H * W, sc)      # unet_causal_3d_blocks.py:884-895
                    st = None
                # the tensor this resnet returns is normalised next (the following resnet's norm1, or conv_norm_out after the last
                # block) unless an upsampler or a t_ops interpolation comes in between: then its conv2 takes the statistics
                feeds_gn = not eia[j] and (j + 1 < n_res and not eib[j + 1] or j + 1 == n_res and not (sp or tm))
                h = self._resnet(P, f"{pre}up_blocks.{i}.resnets.{j}.", h, T, H, W, x_stats=st, out_stats=feeds_gn)
                h, st = h if feeds_gn else (h, None)
                if eia[j]:
                    h, T = V.temporal_nearest_up(h, T, H * W, sc)
            if sp or tm:
                T2, H2, W2 = (1 + 2 * (T - 1) if tm else T), (2 * H if sp else H), (2 * W if sp else W)
                name = f"{pre}up_blocks.{i}.upsamplers.0.conv.conv"
                # the upsampler's output goes straight into the next block's first norm1
                nxt = self._block_cfg(dec_ops.get("up_blocks"), i + 1) or {}
                feeds_gn = i + 1 < nb and not nxt.get("enable_t_interp_before_block", [False])[0]
                if name + "#subpixel" in P:
                    w_sub, table, ntap, b, cip, cop = P[name + "#subpixel"]
                    h = V.conv3d_upsampled_subpixel(h, w_sub, table, ntap, b, T, H, W, cip, cop, tm, gn_stats=feeds_gn)
                else:
                    h = self._conv(P, name, h, T2, H2, W2, up_t=tm, up_hw=sp, gn_stats=feeds_gn)
                h, st = h if feeds_gn else (h, None)
                T, H, W = T2, H2, W2
        if pre + "conv_out.conv#cout4" in P and h.shape[1] == P[pre + "conv_out.conv#cout4"][2]:
            wf, b, ci, co = P[pre + "conv_out.conv#cout4"]
            gw, gb = P[pre + "conv_norm_out"]
            aff = V.groupnorm_affine_from_stats(st, gw, gb, 32, 1e-6) if st is not None else V.groupnorm_affine(h, gw, gb, 32, 1e-6)
            return V.conv_cout4(h, aff, True, wf, b, T, H, W, ci, co), T, H, W
        h = self._gn(P, pre + "conv_norm_out", h, stats=st)
        out = self._conv(P, pre + "conv_out.conv", h, T, H, W)
        return out, T, H, W

    # ------------------------------------------------------------------ shared by both halves: mid block, t_ops bookkeeping
    @staticmethod
    def _block_cfg(cfgs, i):
        for c in cfgs or []:
            if c.get("block_index") == i:
                return c
        return None

    @staticmethod
    def _pool_conf(cfg, n_res, who):
        """apply_t_ops_config / apply_t_ops_config_midblock (unet_causal_3d_blocks.py:622-645,736-762): per-resnet
        (pool before, pool after, kernel, stride); kernel defaults to 2, stride to 2."""
        if not cfg or "enable_t_pool_before_block" not in cfg:
            return [(False, False, 0, 0)] * n_res
        epb, epa = cfg.get("enable_t_pool_before_block", []), cfg.get("enable_t_pool_after_block", [])
        if len(epb) != n_res or len(epa) != n_res:
            raise ValueError(f"[{who}] T-ops config mismatch: we have {n_res} ResnetBlock(s), but got list lengths: "
                             f"{[len(epb), len(epa)]}")
        k, s = int(cfg.get("pool_t_kernel", 2)), int(cfg.get("pool_t_stride", 2))
        return [(bool(epb[i]), bool(epa[i]), k, s) for i in range(n_res)]

    def _mid_block(self, P, pre, h, T, H, W, cfg=None):
        """UNetMidBlockCausal3D.forward (unet_causal_3d_blocks.py:647-674): resnet 0, attention, resnet 1, with the fork's
        temporal pools around each resnet (the attention runs before resnet 1's `before` pool)."""
        conf = self._pool_conf(cfg, 2, "UNetMidBlockCausal3D")
        for i in range(2):
            if i > 0:
                h = self._mid_attention(P, pre + "attentions.0.", h, T, H * W)
            before, after, k, s = conf[i]
            if before:
                h, T = V.temporal_avg_pool(h, T, H * W, k, s)
            h = self._resnet(P, f"{pre}resnets.{i}.", h, T, H, W)
            if after:
                h, T = V.temporal_avg_pool(h, T, H * W, k, s)
        return h, T

    def apply_t_ops_config(self, t_ops_config):
        """The fork's `_apply_t_ops_config_to_vae(vae, cfg)` (vae/__init__.py:15-63): `t_ops_config` is the parsed
        t_ops_config.json (or None to clear).  Validated here so a malformed config fails at load time, like the reference."""
        if t_ops_config is not None:
            n = self.config.layers_per_block
            enc, dec = t_ops_config.get("encoder", {}), t_ops_config.get("decoder", {})
            for bc in enc.get("down_blocks", []):
                self._pool_conf(bc, n, "DownEncoderBlockCausal3D")
            self._pool_conf(enc.get("mid_block"), 2, "UNetMidBlockCausal3D")
            self._pool_conf(dec.get("mid_block"), 2, "UNetMidBlockCausal3D")
            for bc in dec.get("up_blocks", []):
                for key in ("enable_t_interp_before_block", "enable_t_interp_after_block"):
                    if len(bc.get(key, [False] * (n + 1))) != n + 1:
                        raise ValueError(f"[UpDecoderBlockCausal3D] config mismatch: expecting {n + 1} bools in each list.")
        self._t_ops = t_ops_config
        return self

    # ------------------------------------------------------------------ encoder on one video tile
    def _encode_tile(self, x_view: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
        """x_view: [3,T,H,W] strided view (any float dtype) -> (moments, channels-last fp16 [T'*H'*W', 2*latent], T', H', W').
        EncoderCausal3D.forward + quant_conv (vae/vae.py:116-136, autoencoder_kl_causal_3d.py:289-292)."""
        if not self.with_encoder:
            raise RuntimeError("this AutoencoderKLCausal3D was built without the encode half (with_encoder=False)")
        P = self._prepare()
        c, T, H, W = x_view.shape
        x = V.latent_tile(x_view if x_view.dtype == torch.float32 else x_view.to(torch.float32), 64)
        pre = "encoder."
        enc_ops = (self._t_ops or {}).get("encoder", {})
        h = self._conv(P, pre + "conv_in.conv", x, T, H, W)
        boc = self.config.block_out_channels
        nb, n_res = len(boc), self.config.layers_per_block
        for i in range(nb):
            bc = self._block_cfg(enc_ops.get("down_blocks"), i)
            conf = self._pool_conf(bc, n_res, "DownEncoderBlockCausal3D")
            for j in range(n_res):
                before, after, k, s = conf[j]
                if before:
                    h, T = V.temporal_avg_pool(h, T, H * W, k, s)      # unet_causal_3d_blocks.py:764-770
                h = self._resnet(P, f"{pre}down_blocks.{i}.resnets.{j}.", h, T, H, W)
                if after:
                    h, T = V.temporal_avg_pool(h, T, H * W, k, s)
            sp = i < 3
            tm = (i >= nb - 1 - 2) and (i != nb - 1)
            if sp or tm:
                stride = ((2 if tm else 1), (2 if sp else 1), (2 if sp else 1))
                if bc and "downsample_stride" in bc:
                    stride = tuple(int(v) for v in bc["downsample_stride"])      # :737-742
                wt, b, cip, cop = P[f"{pre}down_blocks.{i}.downsamplers.0.conv.conv"]
                h, T, H, W = V.conv3d_causal_strided(self._pad_channels(h, cip), wt, b, T, H, W, cip, cop, stride)
        h, T = self._mid_block(P, pre + "mid_block.", h, T, H, W, enc_ops.get("mid_block"))
        h = self._gn(P, pre + "conv_norm_out", h)
        h = self._conv(P, pre + "conv_out.conv", h, T, H, W)
        wq, bq = P["quant_conv"]
        moments = V.gemm_f16(self._pad_channels(h, wq.shape[1]), wq, bq, k=wq.shape[1])
        return moments, T, H, W

    def _plain_encode(self, x4):
        buf, T, H, W = self._encode_tile(x4)
        c = 2 * self.config.latent_channels
        out = torch.empty(c, T, H, W, dtype=F16, device=buf.device)
        V.copy4d_(self._cl_view(buf, T, H, W, c), out)
        return out

    def _spatial_tiled_encode(self, x4):
        """autoencoder_kl_causal_3d.py:362-420 on a [3,T,H,W] view; returns planar fp16 moments [2*latent,T',H',W']."""
        ov = int(self.tile_sample_min_size * (1 - self.tile_overlap_factor))
        ext = int(self.tile_latent_min_size * self.tile_overlap_factor)
        lim = self.tile_latent_min_size - ext
        c = 2 * self.config.latent_channels
        rows = []
        for i in range(0, x4.shape[-2], ov):
            row = []
            for j in range(0, x4.shape[-1], ov):
                buf, T, H, W = self._encode_tile(x4[:, :, i:i + self.tile_sample_min_size, j:j + self.tile_sample_min_size])
                row.append(self._cl_view(buf, T, H, W, c))
            rows.append(row)
        heights = [min(r[0].shape[2], lim) for r in rows]
        widths = [min(t.shape[3], lim) for t in rows[0]]
        out = torch.empty(c, rows[0][0].shape[1], sum(heights), sum(widths), dtype=F16, device=x4.device)
        y0 = 0
        for i, row in enumerate(rows):
            x0 = 0
            for j, tile in enumerate(row):
                if i > 0:
                    a = rows[i - 1][j]
                    e = min(a.shape[2], tile.shape[2], ext)
                    V.blend_(a[:, :, a.shape[2] - e:, :], tile[:, :, :e, :], 2, e)
                if j > 0:
                    a = row[j - 1]
                    e = min(a.shape[3], tile.shape[3], ext)
                    V.blend_(a[:, :, :, a.shape[3] - e:], tile[:, :, :, :e], 3, e)
                V.copy4d_(tile[:, :, :heights[i], :widths[j]], out[:, :, y0:y0 + heights[i], x0:x0 + widths[j]])
                x0 += widths[j]
            y0 += heights[i]
        return out

    def _temporal_tiled_encode(self, x4):
        """autoencoder_kl_causal_3d.py:470-510."""
        T = x4.shape[1]
        ov = int(self.tile_sample_min_tsize * (1 - self.tile_overlap_factor))
        ext = int(self.tile_latent_min_tsize * self.tile_overlap_factor)
        lim = self.tile_latent_min_tsize - ext
        row = []
        for i in range(0, T, ov):
            tile = x4[:, i:i + self.tile_sample_min_tsize + 1]
            if self.use_spatial_tiling and (tile.shape[-1] > self.tile_sample_min_size or tile.shape[-2] > self.tile_sample_min_size):
                enc = self._spatial_tiled_encode(tile)
            else:
                enc = self._plain_encode(tile)
            if i > 0:
                enc = enc[:, 1:]
            row.append(enc)
        lens = [min(t.shape[1], lim + (1 if i == 0 else 0)) for i, t in enumerate(row)]
        out = torch.empty(row[0].shape[0], sum(lens), row[0].shape[2], row[0].shape[3], dtype=F16, device=x4.device)
        t0 = 0
        for i, tile in enumerate(row):
            if i > 0:
                a = row[i - 1]
                e = min(a.shape[1], tile.shape[1], ext)
                if e > 0:
                    V.blend_(a[:, a.shape[1] - e:], tile[:, :e], 1, e)
            if lens[i] > 0:
                V.copy4d_(tile[:, :lens[i]], out[:, t0:t0 + lens[i]])
            t0 += lens[i]
        return out

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """autoencoder_kl_causal_3d.py:259-296: x [B,3,T,H,W] -> AutoencoderKLOutput(latent_dist) / (posterior,)."""
        assert len(x.shape) == 5, "The input tensor should have 5 dimensions."
        if x.shape[0] != 1:
            moments = torch.cat([self._encode_moments(xs) for xs in x.split(1)])
        else:
            moments = self._encode_moments(x)
        posterior = DiagonalGaussianDistribution(moments)
        if not return_dict:
            return (posterior,)
        return SimpleNamespace(latent_dist=posterior, tiles_ci=None)

    def _encode_moments(self, x):
        x4 = x[0]
        if self._t_ops is not None and (self.use_temporal_tiling or self.use_spatial_tiling):
            raise NotImplementedError("t_ops change the compression ratios the tile/blend geometry assumes; the fork runs them untiled")
        if self.use_temporal_tiling and x4.shape[1] > self.tile_sample_min_tsize:
            return self._temporal_tiled_encode(x4)[None]
        if self.use_spatial_tiling and (x4.shape[-1] > self.tile_sample_min_size or x4.shape[-2] > self.tile_sample_min_size):
            return self._spatial_tiled_encode(x4)[None]
        return self._plain_encode(x4)[None]

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                return_posterior: bool = False, generator: Optional[torch.Generator] = None):
        """autoencoder_kl_causal_3d.py:545-580 (what the fork's infer.py:52-57 calls): encode -> sample()/mode() -> decode."""
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        dec = self.decode(z).sample
        if not return_dict:
            return (dec, posterior) if return_posterior else (dec,)
        return SimpleNamespace(sample=dec, posterior=posterior) if return_posterior else SimpleNamespace(sample=dec)

    # ------------------------------------------------------------------ tiling (reference loop order)
    @staticmethod
    def _cl_view(buf, T, H, W, c=3):
        """[C,T,H,W] strided view of a channels-last [T*H*W, 8] buffer."""
        return buf.as_strided((c, T, H, W), (1, H * W * buf.stride(0), W * buf.stride(0), buf.stride(0)), buf.storage_offset())

    def _plain_decode(self, z4):
        buf, T, H, W = self._take_tile(z4)
        out = torch.empty(3, T, H, W, dtype=F16, device=buf.device)
        V.copy4d_(self._cl_view(buf, T, H, W), out)
        return out

    # ------------------------------------------------------------------ tile parallelism (new vs the reference,
    # which decodes every tile on every rank).  Tiles are independent until the blend: each rank decodes its share, the decoded
    # tiles are all-gathered (one RCCL all-gather per round of P tiles, in flight while the next round decodes), and every rank
    # runs the reference's blend loops over the complete set, so decode() still returns the whole video on every rank.
    def enable_tile_parallel(self, group=None, enable: bool = True, gather: Optional[str] = None):
        """gather (or HV_VAE_TILE_GATHER): "all" (default) - every decoded tile is all-gathered, every rank blends and returns the
        whole video (what the reference's callers get, since it decodes everything everywhere); "rank0" - tiles are gathered to
        group rank 0 only, which blends and returns the video; the other ranks return zeros of the video's shape (the reference's
        driver saves on rank 0 only: sample_video.py).  1/P of the receive traffic per rank; never timed on hardware - both stay
        selectable so the first multi-GPU run can A/B them."""
        self._tp_enabled = enable
        self._tp_group = group
        import os
        g = (gather or os.environ.get("HV_VAE_TILE_GATHER", "all")).lower()
        if g not in ("all", "rank0"):
            raise ValueError(f"tile gather mode must be 'all' or 'rank0', got {g!r}")
        self._tp_gather = g

    def _spatial_views(self, z4):
        ov = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        for i in range(0, z4.shape[-2], ov):
            for j in range(0, z4.shape[-1], ov):
                yield z4[:, :, i:i + self.tile_latent_min_size, j:j + self.tile_latent_min_size]

    def _needs_spatial(self, z4):
        return self.use_spatial_tiling and (z4.shape[-1] > self.tile_latent_min_size or z4.shape[-2] > self.tile_latent_min_size)

    def _tile_views(self, z4):
        """Latent views of every tile, in exactly the order the decode loops below consume them."""
        if self.use_temporal_tiling and z4.shape[1] > self.tile_latent_min_tsize:
            ov = int(self.tile_latent_min_tsize * (1 - self.tile_overlap_factor))
            for i in range(0, z4.shape[1], ov):
                tile = z4[:, i:i + self.tile_latent_min_tsize + 1]
                if self._needs_spatial(tile):
                    yield from self._spatial_views(tile)
                else:
                    yield tile
        elif self._needs_spatial(z4):
            yield from self._spatial_views(z4)
        else:
            yield z4

    @staticmethod
    def _assign_tiles(costs: List[int], world: int) -> List[List[int]]:
        """Longest-processing-time-first: tiles in descending cost, each to the least-loaded rank (ties -> lowest rank).
        Deterministic, so every rank derives the same plan without communicating."""
        load = [0] * world
        plan: List[List[int]] = [[] for _ in range(world)]
        for k in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
            r = min(range(world), key=lambda r: (load[r], r))
            plan[r].append(k)
            load[r] += costs[k]
        return plan

    def _decode_tiles_sharded(self, z4, group):
        dist = _dist()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        views = list(self._tile_views(z4))
        tc = self.time_compression_ratio
        dims = [((v.shape[1] - 1) * tc + 1, v.shape[2] * 8, v.shape[3] * 8) for v in views]     # decoded T', H', W'
        plan = self._assign_tiles([v.shape[1] * v.shape[2] * v.shape[3] for v in views], world)
        rounds = max(len(p) for p in plan)
        max_rows = max(t * h * w for t, h, w in dims)
        to_root = getattr(self, "_tp_gather", "all") == "rank0"
        root = dist.get_global_rank(group, 0) if group is not None else 0
        gathered = torch.empty(rounds, world, max_rows, 8, dtype=F16, device=z4.device) if (rank == 0 or not to_root) else None
        mine = torch.zeros(rounds, max_rows, 8, dtype=F16, device=z4.device)
        works = []
        for r in range(rounds):
            if r < len(plan[rank]):
                k = plan[rank][r]
                buf, T, H, W = self._decode_tile(views[k])
                assert (T, H, W) == dims[k] and buf.shape[1] == 8
                mine[r, :buf.shape[0]].copy_(buf)
            if to_root:
                works.append(dist.gather(mine[r], [gathered[r, p] for p in range(world)] if rank == 0 else None, dst=root, group=group,
                                         async_op=True))
            else:
                works.append(dist.all_gather_into_tensor(gathered[r].view(world * max_rows, 8), mine[r], group=group, async_op=True))
        for w in works:
            w.wait()
        if gathered is None:
            return None          # "rank0" mode, not the root: nothing to blend here
        out = [None] * len(views)
        for p in range(world):
            for r, k in enumerate(plan[p]):
                t, h, w = dims[k]
                out[k] = (gathered[r, p, :t * h * w], t, h, w)
        return out

    def _decode_tiles_concurrent(self, z4):
        """All tiles of a tiled decode, decoded on `decode_streams` HIP streams at once (greedy by tile size), returned in the order
        the blend loops consume them.  A tile is a chain of ~150 dependent launches whose grids often end in a partly filled last
        round of workgroups (one per CU: 544 workgroups = 2.1 rounds on 256 CUs), and of small kernels (statistics folds, the
        mid-block attention); a second, independent tile fills those gaps.  Same kernels on the same data: bit-identical output."""
        views = list(self._tile_views(z4))
        n = int(self.decode_streams)
        if n < 2 or len(views) < 2 or not z4.is_cuda:
            return None
        dev = z4.device
        if self._streams is None or len(self._streams) != n or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        self._prepare()                                   # weight preparation on the caller's stream, before the fork
        cur = torch.cuda.current_stream(dev)
        for st in self._streams:
            st.wait_stream(cur)
        load = [0] * n
        out = [None] * len(views)
        for k, v in enumerate(views):
            i = min(range(n), key=lambda i: (load[i], i))
            load[i] += v.shape[1] * v.shape[2] * v.shape[3]
            with torch.cuda.stream(self._streams[i]):
                buf, T, H, W = self._decode_tile(v)
            buf.record_stream(cur)                        # consumed by the blend / copy kernels on the caller's stream
            out[k] = (buf, T, H, W)
        for st in self._streams:
            cur.wait_stream(st)
        return out

    def _take_tile(self, z_view):
        """The decode loops' tile source: decode here, or (tile-parallel) the next pre-decoded tile."""
        q = getattr(self, "_tile_queue", None)
        if q is None:
            return self._decode_tile(z_view)
        buf, T, H, W = q.pop(0)
        assert (T, H, W) == ((z_view.shape[1] - 1) * self.time_compression_ratio + 1, z_view.shape[2] * 8, z_view.shape[3] * 8)
        return buf, T, H, W

    def _spatial_tiled_decode(self, z4):
        """autoencoder_kl_causal_3d.py:422-469 on a [C,T,H,W] view; returns planar fp16 [3,T',H',W']."""
        ov = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        ext = int(self.tile_sample_min_size * self.tile_overlap_factor)
        lim = self.tile_sample_min_size - ext
        rows = []
        for i in range(0, z4.shape[-2], ov):
            row = []
            for j in range(0, z4.shape[-1], ov):
                buf, T, H, W = self._take_tile(z4[:, :, i:i + self.tile_latent_min_size, j:j + self.tile_latent_min_size])
                row.append(self._cl_view(buf, T, H, W))
            rows.append(row)
        heights = [min(r[0].shape[2], lim) for r in rows]
        widths = [min(t.shape[3], lim) for t in rows[0]]
        T = rows[0][0].shape[1]
        out = torch.empty(3, T, sum(heights), sum(widths), dtype=F16, device=z4.device)
        y0 = 0
        for i, row in enumerate(rows):
            x0 = 0
            for j, tile in enumerate(row):
                if i > 0:
                    a = rows[i - 1][j]
                    e = min(a.shape[2], tile.shape[2], ext)
                    V.blend_(a[:, :, a.shape[2] - e:, :], tile[:, :, :e, :], 2, e)
                if j > 0:
                    a = row[j - 1]
                    e = min(a.shape[3], tile.shape[3], ext)
                    V.blend_(a[:, :, :, a.shape[3] - e:], tile[:, :, :, :e], 3, e)
                V.copy4d_(tile[:, :, :heights[i], :widths[j]], out[:, :, y0:y0 + heights[i], x0:x0 + widths[j]])
                x0 += widths[j]
            y0 += heights[i]
        return out

    def _temporal_tiled_decode(self, z4):
        """autoencoder_kl_causal_3d.py:510-541."""
        T = z4.shape[1]
        ov = int(self.tile_latent_min_tsize * (1 - self.tile_overlap_factor))
        ext = int(self.tile_sample_min_tsize * self.tile_overlap_factor)
        lim = self.tile_sample_min_tsize - ext
        row = []
        for i in range(0, T, ov):
            tile = z4[:, i:i + self.tile_latent_min_tsize + 1]
            if self.use_spatial_tiling and (tile.shape[-1] > self.tile_latent_min_size or tile.shape[-2] > self.tile_latent_min_size):
                dec = self._spatial_tiled_decode(tile)
            else:
                dec = self._plain_decode(tile)
            if i > 0:
                dec = dec[:, 1:]
            row.append(dec)
        lens = [min(t.shape[1], lim + (1 if i == 0 else 0)) for i, t in enumerate(row)]
        out = torch.empty(3, sum(lens), row[0].shape[2], row[0].shape[3], dtype=F16, device=z4.device)
        t0 = 0
        for i, tile in enumerate(row):
            if i > 0:
                a = row[i - 1]
                e = min(a.shape[1], tile.shape[1], ext)
                V.blend_(a[:, a.shape[1] - e:], tile[:, :e], 1, e)
            V.copy4d_(tile[:, :lens[i]], out[:, t0:t0 + lens[i]])
            t0 += lens[i]
        return out

    def _decode(self, z: torch.Tensor):
        assert len(z.shape) == 5, "The input tensor should have 5 dimensions."
        if z.shape[0] != 1:
            raise NotImplementedError("batch 1 (use_slicing splits larger batches)")
        z4 = z[0].to(torch.float32)
        self._tile_queue = None
        if self._t_ops is not None and (self.use_temporal_tiling or self.use_spatial_tiling):
            raise NotImplementedError("t_ops change the compression ratios the tile/blend geometry assumes; the fork runs them untiled")
        if getattr(self, "_tp_enabled", False):
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self._tp_group) > 1:
                self._tile_queue = self._decode_tiles_sharded(z4, self._tp_group)
                if self._tile_queue is None:
                    # gather-to-rank-0 mode on a non-root rank: the video lives on rank 0; same shape, zeros, here
                    tc = self.time_compression_ratio
                    return torch.zeros(1, 3, (z4.shape[1] - 1) * tc + 1, z4.shape[2] * 8, z4.shape[3] * 8, dtype=F16, device=z4.device)
        if self._tile_queue is None and (self.use_temporal_tiling or self.use_spatial_tiling):
            self._tile_queue = self._decode_tiles_concurrent(z4)
        try:
            return self._decode_assembled(z4)
        finally:
            self._tile_queue = None

    def _decode_assembled(self, z4):
        if self.use_temporal_tiling and z4.shape[1] > self.tile_latent_min_tsize:
            return self._temporal_tiled_decode(z4)[None]
        if self.use_spatial_tiling and (z4.shape[-1] > self.tile_latent_min_size or z4.shape[-2] > self.tile_latent_min_size):
            return self._spatial_tiled_decode(z4)[None]
        return self._plain_decode(z4)[None]

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        """autoencoder_kl_causal_3d.py:316-342: returns (sample,) or an object with .sample; sample fp16 [B,3,T,H,W]."""
        if self.use_slicing and z.shape[0] > 1:
            decoded = torch.cat([self._decode(zs) for zs in z.split(1)])
        else:
            decoded = self._decode(z)
        if not return_dict:
            return (decoded,)
        return SimpleNamespace(sample=decoded)
