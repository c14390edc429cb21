"""Parameter holders whose attribute names reproduce the reference state-dict keys
(hyvideo/modules/{modulate_layers,mlp_layers,embed_layers,norm_layers}.py) plus the few-launch `run`
methods of the small `vec`-side networks.  No torch arithmetic: `run` only calls ops.* kernels."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops

BF16 = torch.bfloat16


class ParamLinear(nn.Module):
    """weight [out,in] (+ bias [out]) - the parameters of an nn.Linear."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype), requires_grad=False)
        else:
            self.register_parameter("bias", None)

    _fp8_scratch = {}

    def w(self) -> torch.Tensor:
        """The bf16 weight the GEMM consumes.  For FP8-converted layers (fp8_optimization.convert_fp8_linear) the
        e4m3fn weight is dequantised per call into one shared scratch buffer, W = bf16(bf16(w8) * fp8_scale), exactly the
        reference's per-forward dequantisation (fp8_optimization.py:50-53,69-75); launches are stream-ordered, so a
        single scratch per device is enough and the bf16 copy of the model never exists in HBM."""
        wt = self.weight
        if wt.dtype != torch.float8_e4m3fn:
            return wt
        key = str(wt.device)
        buf = ParamLinear._fp8_scratch.get(key)
        if buf is None or buf.numel() < wt.numel():
            buf = torch.empty(wt.numel(), dtype=BF16, device=wt.device)
            ParamLinear._fp8_scratch[key] = buf
        out = buf[: wt.numel()].view(wt.shape)
        ops.fp8_dequant(wt, self.fp8_scale, out)
        return out


class ParamNormWeight(nn.Module):
    """RMSNorm gain (norm_layers.py:30-31)."""

    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)


class ParamLayerNorm(nn.Module):
    """Affine LayerNorm parameters (token_refiner.py:33-35,56-58)."""

    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype), requires_grad=False)


class ModulateDiT(nn.Module):
    """modulate_layers.py:7-28: `linear` [factor*d, d]; applied as Linear(SiLU(vec)) by the blocks."""

    def __init__(self, hidden_size: int, factor: int, device=None, dtype=None):
        super().__init__()
        self.linear = ParamLinear(hidden_size, factor * hidden_size, True, device, dtype)


class MLP(nn.Module):
    """mlp_layers.py:13-59 (fc1 -> act -> fc2; dropout 0, norm Identity)."""

    def __init__(self, in_channels: int, hidden_channels: int, device=None, dtype=None):
        super().__init__()
        self.fc1 = ParamLinear(in_channels, hidden_channels, True, device, dtype)
        self.fc2 = ParamLinear(hidden_channels, in_channels, True, device, dtype)


class MLPEmbedder(nn.Module):
    """mlp_layers.py:63-73: out_layer(silu(in_layer(x)))."""

    def __init__(self, in_dim: int, hidden_dim: int, device=None, dtype=None):
        super().__init__()
        self.in_layer = ParamLinear(in_dim, hidden_dim, True, device, dtype)
        self.out_layer = ParamLinear(hidden_dim, hidden_dim, True, device, dtype)

    def run(self, x_bf16: torch.Tensor, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = ops.linear_smallm(x_bf16, self.in_layer.weight, self.in_layer.bias, silu_out=True)
        return ops.linear_smallm(h, self.out_layer.weight, self.out_layer.bias, addend=addend)


class _Seq(nn.Module):
    """children named by integer strings, like nn.Sequential indices ("mlp.0", "mlp.2", "adaLN_modulation.1")."""

    def __init__(self, **mods):
        super().__init__()
        for k, m in mods.items():
            self.add_module(k.lstrip("_"), m)

    def __getitem__(self, i):
        return getattr(self, str(i))


class TimestepEmbedder(nn.Module):
    """embed_layers.py:120-157: sinusoid(256) -> Linear -> SiLU -> Linear."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256, max_period: float = 10000.0,
                 device=None, dtype=None):
        super().__init__()
        self.frequency_embedding_size = frequency_embedding_size
        self.max_period = max_period
        self.mlp = _Seq(_0=ParamLinear(frequency_embedding_size, hidden_size, True, device, dtype),
                        _2=ParamLinear(hidden_size, hidden_size, True, device, dtype))

    def run(self, t_f32: torch.Tensor, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
        f = ops.timestep_embedding(t_f32, self.frequency_embedding_size, self.max_period)
        h = ops.linear_smallm(f, self.mlp[0].weight, self.mlp[0].bias, silu_out=True)
        return ops.linear_smallm(h, self.mlp[2].weight, self.mlp[2].bias, addend=addend)


class _ConvParams(nn.Module):
    def __init__(self, in_chans, embed_dim, patch_size, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(embed_dim, in_chans, *patch_size, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(embed_dim, device=device, dtype=dtype), requires_grad=False)


class PatchEmbed(nn.Module):
    """embed_layers.py:9-59: Conv3d(k = s = patch) + flatten, executed as gather (hv_patchify) + K=64 MFMA GEMM."""

    def __init__(self, patch_size, in_chans: int, embed_dim: int, device=None, dtype=None):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.proj = _ConvParams(in_chans, embed_dim, self.patch_size, device, dtype)

    def run(self, x_f32: torch.Tensor, ws, s_img: int):
        c = x_f32.shape[0]
        if ws.patches is None or ws.patches.shape != (s_img, c * 4):
            ws.patches = torch.empty(s_img, c * 4, dtype=BF16, device=x_f32.device)
        ops.patchify(x_f32, out=ws.patches)
        w = self.proj.weight.reshape(self.proj.weight.shape[0], -1)  # [d, C*1*2*2], k index = c*4 + ph*2 + pw
        ops.gemm(ws.patches, w, self.proj.bias, out=ws.x[:s_img])


class FinalLayer(nn.Module):
    """mlp_layers.py:76-118: LN -> modulate(shift, scale from adaLN(SiLU(vec))) -> Linear(d -> pt*ph*pw*C).
    The reference builds `linear` WITHOUT dtype when patch_size is a list (:96-101), so checkpoints/`.to()` may
    leave it fp32; autocast computes it in bf16: it is cast once and cached here."""

    def __init__(self, hidden_size: int, patch_size, out_channels: int, device=None, dtype=None):
        super().__init__()
        n_out = patch_size[0] * patch_size[1] * patch_size[2] * out_channels
        self.linear = ParamLinear(hidden_size, n_out, True, device, dtype)
        self.adaLN_modulation = _Seq(_1=ParamLinear(hidden_size, 2 * hidden_size, True, device, dtype))
        self._w16 = None

    def _bf16_linear(self):
        w, b = self.linear.weight, self.linear.bias
        if w.dtype == BF16:
            return w, b
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if self._w16 is None or self._w16[0] != key:
            self._w16 = (key, w.detach().to(BF16), b.detach().to(BF16))
        return self._w16[1], self._w16[2]

    def run(self, ws, s_img: int, vec: torch.Tensor) -> torch.Tensor:
        d = vec.shape[-1]
        lin = self.adaLN_modulation[1]
        m = ops.linear_smallm(vec, lin.weight, lin.bias, silu_in=True)
        shift, scale = m[0, :d], m[0, d:]
        ops.ln_modulate(ws.x[:s_img], shift, scale, out=ws.xmod[:s_img])
        w, b = self._bf16_linear()
        n_out = w.shape[0]
        if ws.final is None or ws.final.shape != (s_img, n_out):
            ws.final = torch.empty(s_img, n_out, dtype=BF16, device=vec.device)
        return ops.gemm(ws.xmod[:s_img], w, b, out=ws.final)
