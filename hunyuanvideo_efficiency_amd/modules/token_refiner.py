"""SingleTokenRefiner (hyvideo/modules/token_refiner.py:164-236) on the gfx950 kernels: same state-dict keys
(input_embedder, t_embedder, c_embedder, individual_token_refiner.blocks.N.*).  256 text tokens, runs every
denoise step as in the reference (models.py:638)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .layers import MLP, ParamLayerNorm, ParamLinear, TimestepEmbedder, _Seq

BF16 = torch.bfloat16


class TextProjection(nn.Module):
    """embed_layers.py:62-90: linear_2(silu(linear_1(x)))."""

    def __init__(self, in_channels: int, hidden_size: int, device=None, dtype=None):
        super().__init__()
        self.linear_1 = ParamLinear(in_channels, hidden_size, True, device, dtype)
        self.linear_2 = ParamLinear(hidden_size, hidden_size, True, device, dtype)

    def run(self, x_bf16: torch.Tensor, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = ops.linear_smallm(x_bf16, self.linear_1.weight, self.linear_1.bias, silu_out=True)
        return ops.linear_smallm(h, self.linear_2.weight, self.linear_2.bias, addend=addend)


class IndividualTokenRefinerBlock(nn.Module):
    """token_refiner.py:16-100 (qk_norm=False -> Identity: no q/k norm parameters)."""

    def __init__(self, hidden_size: int, heads_num: int, mlp_width_ratio: float = 4.0, device=None, dtype=None):
        super().__init__()
        self.heads_num = heads_num
        self.norm1 = ParamLayerNorm(hidden_size, device, dtype)
        self.self_attn_qkv = ParamLinear(hidden_size, hidden_size * 3, True, device, dtype)
        self.self_attn_proj = ParamLinear(hidden_size, hidden_size, True, device, dtype)
        self.norm2 = ParamLayerNorm(hidden_size, device, dtype)
        self.mlp = MLP(hidden_size, int(hidden_size * mlp_width_ratio), device, dtype)
        self.adaLN_modulation = _Seq(_1=ParamLinear(hidden_size, 2 * hidden_size, True, device, dtype))


class IndividualTokenRefiner(nn.Module):
    def __init__(self, hidden_size: int, heads_num: int, depth: int, mlp_width_ratio: float = 4.0, device=None, dtype=None):
        super().__init__()
        self.blocks = nn.ModuleList([IndividualTokenRefinerBlock(hidden_size, heads_num, mlp_width_ratio, device, dtype)
                                     for _ in range(depth)])


class SingleTokenRefiner(nn.Module):
    def __init__(self, in_channels: int, hidden_size: int, heads_num: int, depth: int, mlp_width_ratio: float = 4.0,
                 mlp_drop_rate: float = 0.0, act_type: str = "silu", qk_norm: bool = False, qk_norm_type: str = "layer",
                 qkv_bias: bool = True, attn_mode: str = "torch", dtype=None, device=None):
        super().__init__()
        assert attn_mode == "torch", "Only support 'torch' mode for token refiner."
        if act_type != "silu" or qk_norm or not qkv_bias or mlp_drop_rate != 0.0:
            raise NotImplementedError("token refiner kernels: SiLU MLP, no qk-norm, qkv bias (the shipped configuration)")
        self.hidden_size, self.heads_num = hidden_size, heads_num
        self.input_embedder = ParamLinear(in_channels, hidden_size, True, device, dtype)
        self.t_embedder = TimestepEmbedder(hidden_size, device=device, dtype=dtype)
        self.c_embedder = TextProjection(in_channels, hidden_size, device, dtype)
        self.individual_token_refiner = IndividualTokenRefiner(hidden_size, heads_num, depth, mlp_width_ratio, device, dtype)
        self._bufs = None

    def run(self, text: torch.Tensor, t_f32: torch.Tensor, mask: Optional[torch.Tensor], out: torch.Tensor,
            cache: Optional[dict] = None):
        """text [L, in_channels] bf16, mask [1, L] (prefix mask) or None; writes the refined tokens into out [L, d].
        `cache` (a dict the caller keeps per prompt): the reference re-runs the whole refiner every step (models.py:638) although
        only `t` changes; the timestep-INDEPENDENT prefix - input_embedder(text) and the masked mean that feeds c_embedder - is
        computed once per prompt and reused (the two refiner blocks are modulated by t and run every step)."""
        from .attenion import n_valid_text
        L, d, H = text.shape[0], self.hidden_size, self.heads_num
        dev = text.device
        n_valid = L if mask is None else n_valid_text(mask)
        hit = cache is not None and "emb" in cache and cache["emb"].shape == (L, d) and cache["emb"].device == dev
        if hit:
            mask_i32, ctx = cache["mask_i32"], cache["ctx"]
        else:
            mask_i32 = None if mask is None else mask[0].to(device=dev, dtype=torch.int32).contiguous()
            ctx = ops.masked_mean(text, mask_i32)
        # c = t_embedder(t) + c_embedder(masked mean of the raw text states)      (token_refiner.py:220-229)
        t_aware = self.t_embedder.run(t_f32)
        c = self.c_embedder.run(ctx.reshape(1, -1), addend=t_aware)
        if self._bufs is None or self._bufs[0].shape[0] != L or self._bufs[0].device != dev:
            self._bufs = (torch.empty(L, d, dtype=BF16, device=dev), torch.empty(L, 3 * d, dtype=BF16, device=dev),
                          torch.empty(L, d, dtype=BF16, device=dev), torch.empty(L, 4 * d, dtype=BF16, device=dev))
        norm, qkv, attn, hid = self._bufs
        mlp_hidden = self.individual_token_refiner.blocks[0].mlp.fc1.weight.shape[0]
        if hid.shape[1] != mlp_hidden:
            hid = torch.empty(L, mlp_hidden, dtype=BF16, device=dev)
            self._bufs = (norm, qkv, attn, hid)
        x = out
        if hit:
            x.copy_(cache["emb"])
        else:
            ops.gemm(text, self.input_embedder.weight, self.input_embedder.bias, out=x)
            if cache is not None:
                cache.update(emb=x.clone(), ctx=ctx, mask_i32=mask_i32)
        for blk in self.individual_token_refiner.blocks:
            ada = blk.adaLN_modulation[1]
            g = ops.linear_smallm(c, ada.weight, ada.bias, silu_in=True)
            gate_msa, gate_mlp = g[0, :d], g[0, d:]
            ops.ln_modulate(x, blk.norm1.bias, blk.norm1.weight, out=norm, affine=True)
            ops.gemm(norm, blk.self_attn_qkv.weight, blk.self_attn_qkv.bias, out=qkv)
            # boolean mask = mask_i & mask_j with column 0 forced True (token_refiner.py:143-157):
            # valid rows attend the valid prefix; fully masked rows see only key 0 -> softmax is one-hot -> v[0]
            if n_valid > 0:
                ops.attn_fwd(qkv[:n_valid, :d], qkv[:n_valid, d:2 * d], qkv[:n_valid, 2 * d:], attn[:n_valid], H)
            if n_valid < L:
                ops.broadcast_row_(qkv[0, 2 * d:], attn[n_valid:])
            ops.gemm(attn, blk.self_attn_proj.weight, blk.self_attn_proj.bias, out=x, gate=gate_msa, res=x)
            ops.ln_modulate(x, blk.norm2.bias, blk.norm2.weight, out=norm, affine=True)
            ops.gemm(norm, blk.mlp.fc1.weight, blk.mlp.fc1.bias, out=hid, act=ops.ACT_SILU)
            ops.gemm(hid, blk.mlp.fc2.weight, blk.mlp.fc2.bias, out=x, gate=gate_mlp, res=x)
        return x
