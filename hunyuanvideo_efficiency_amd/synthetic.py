"""Deterministic synthetic weights and inputs (no checkpoints, no RNG state).

Every tensor is a pure function of (key name, element index, seed): a 32-bit
integer hash evaluated with int64 tensor arithmetic, so numpy, torch-CPU and
torch-on-GPU produce bit-identical values.  Used for random-init benchmarks and
for the golden fixtures (tools/make_golden.py) so that no weight blobs are
committed.

The key/shape table below restates the state-dict layout of the reference
modules (hyvideo/modules/models.py:48-123,287-317,502-581,
token_refiner.py:33-75,183-209, embed_layers.py:140-150, mlp_layers.py:65-69,
88-112); tools/make_golden.py checks it with ``load_state_dict(strict=True)``
against the imported reference.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch

_M32 = 0xFFFFFFFF


def _hash32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """lowbias32-style avalanche of (idx + seed); idx is int64, result in [0, 2^32)."""
    h = (idx + (seed & _M32)) & _M32
    h = h ^ (h >> 16)
    h = (h * 0x7FEB352D) & _M32
    h = h ^ (h >> 15)
    h = (h * 0x846CA68B) & _M32
    h = h ^ (h >> 16)
    return h


def hashed_uniform(shape, key: str, seed: int = 0, device="cpu", chunk: int = 1 << 26) -> torch.Tensor:
    """fp32 tensor of `shape`, uniform in [-1, 1), a pure function of (key, seed, index)."""
    n = 1
    for s in shape:
        n *= int(s)
    kseed = (zlib.crc32(key.encode()) * 0x9E3779B1 + seed * 0x85EBCA6B + 0x1234567) & _M32
    out = torch.empty(n, dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = _hash32(idx, kseed)
        out[lo:hi] = h.to(torch.float32) * (2.0 / 4294967296.0) - 1.0
    return out.reshape(*shape)


@dataclass
class DiTConfig:
    """Constructor arguments of HYVideoDiffusionTransformer (models.py:449-470)."""
    hidden_size: int = 3072
    heads_num: int = 24
    mlp_width_ratio: float = 4.0
    mm_double_blocks_depth: int = 20
    mm_single_blocks_depth: int = 40
    rope_dim_list: List[int] = field(default_factory=lambda: [16, 56, 56])
    in_channels: int = 16
    out_channels: int = 16
    patch_size: List[int] = field(default_factory=lambda: [1, 2, 2])
    text_states_dim: int = 4096
    text_states_dim_2: int = 768
    guidance_embed: bool = True
    refiner_depth: int = 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.heads_num

    @property
    def mlp_hidden(self) -> int:
        return int(self.hidden_size * self.mlp_width_ratio)


def tiny_config() -> DiTConfig:
    """BASELINE.json configs[0]: tiny DiT, d=256 (heads 2 x 128), 2 blocks (1 double + 1 single)."""
    return DiTConfig(hidden_size=256, heads_num=2, mm_double_blocks_depth=1, mm_single_blocks_depth=1)


def dit_param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    d, hd, mh = cfg.hidden_size, cfg.head_dim, cfg.mlp_hidden
    pt, ph, pw = cfg.patch_size
    t: Dict[str, Tuple[int, ...]] = {}

    def lin(name, n_out, n_in, bias=True):
        t[name + ".weight"] = (n_out, n_in)
        if bias:
            t[name + ".bias"] = (n_out,)

    t["img_in.proj.weight"] = (d, cfg.in_channels, pt, ph, pw)
    t["img_in.proj.bias"] = (d,)
    lin("txt_in.input_embedder", d, cfg.text_states_dim)
    lin("txt_in.t_embedder.mlp.0", d, 256)
    lin("txt_in.t_embedder.mlp.2", d, d)
    lin("txt_in.c_embedder.linear_1", d, cfg.text_states_dim)
    lin("txt_in.c_embedder.linear_2", d, d)
    for i in range(cfg.refiner_depth):
        p = f"txt_in.individual_token_refiner.blocks.{i}."
        t[p + "norm1.weight"] = (d,)
        t[p + "norm1.bias"] = (d,)
        lin(p + "self_attn_qkv", 3 * d, d)
        lin(p + "self_attn_proj", d, d)
        t[p + "norm2.weight"] = (d,)
        t[p + "norm2.bias"] = (d,)
        lin(p + "mlp.fc1", mh, d)
        lin(p + "mlp.fc2", d, mh)
        lin(p + "adaLN_modulation.1", 2 * d, d)
    lin("time_in.mlp.0", d, 256)
    lin("time_in.mlp.2", d, d)
    lin("vector_in.in_layer", d, cfg.text_states_dim_2)
    lin("vector_in.out_layer", d, d)
    if cfg.guidance_embed:
        lin("guidance_in.mlp.0", d, 256)
        lin("guidance_in.mlp.2", d, d)
    for i in range(cfg.mm_double_blocks_depth):
        for s in ("img", "txt"):
            p = f"double_blocks.{i}.{s}_"
            lin(p + "mod.linear", 6 * d, d)
            lin(p + "attn_qkv", 3 * d, d)
            t[p + "attn_q_norm.weight"] = (hd,)
            t[p + "attn_k_norm.weight"] = (hd,)
            lin(p + "attn_proj", d, d)
            lin(p + "mlp.fc1", mh, d)
            lin(p + "mlp.fc2", d, mh)
    for i in range(cfg.mm_single_blocks_depth):
        p = f"single_blocks.{i}."
        lin(p + "linear1", 3 * d + mh, d)
        lin(p + "linear2", d, d + mh)
        t[p + "q_norm.weight"] = (hd,)
        t[p + "k_norm.weight"] = (hd,)
        lin(p + "modulation.linear", 3 * d, d)
    lin("final_layer.linear", pt * ph * pw * cfg.out_channels, d)
    lin("final_layer.adaLN_modulation.1", 2 * d, d)
    return t


def synth_param(key: str, shape, seed: int = 0, device="cpu") -> torch.Tensor:
    """One synthetic parameter (fp32). Matrices: std 0.7/sqrt(fan_in); norm gains: 1+0.1u;
    biases / modulation outputs: small non-zero (the reference zero-inits modulation and the
    final layer, modulate_layers.py:23-25, mlp_layers.py:102-112 - a benchmark on zeros would
    make every block the identity)."""
    u = hashed_uniform(shape, key, seed, device)
    if key.endswith(".weight") and len(shape) == 1:
        return 1.0 + 0.1 * u
    if key.endswith(".bias"):
        return 0.05 * u
    fan_in = 1
    for s in shape[1:]:
        fan_in *= int(s)
    return u * (0.7 * math.sqrt(3.0) / math.sqrt(fan_in))


def synth_dit_state_dict(cfg: DiTConfig, seed: int = 0, device="cpu", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: synth_param(k, shp, seed, device).to(dtype) for k, shp in dit_param_shapes(cfg).items()}


def synth_dit_inputs(cfg: DiTConfig, latent_thw, txt_len: int = 256, n_valid_txt: int = 11,
                     seed: int = 0, device="cpu"):
    """Synthetic step inputs of SURVEY.md 8(d): latents, text states, mask (first n valid), pooled
    vector.  Latents/text are uniform scaled to unit variance (a pure function of the seed)."""
    T, H, W = latent_thw
    s3 = math.sqrt(3.0)
    x = hashed_uniform((1, cfg.in_channels, T, H, W), "latents", seed, device) * s3
    text_states = hashed_uniform((1, txt_len, cfg.text_states_dim), "text_states", seed, device) * s3
    text_states_2 = hashed_uniform((1, cfg.text_states_dim_2), "text_states_2", seed, device) * s3
    text_mask = torch.zeros(1, txt_len, dtype=torch.int64, device=device)
    text_mask[:, :n_valid_txt] = 1
    return x, text_states, text_mask, text_states_2


# ------------------------------------------------------------------------------------------------ VAE decoder
VAE_BLOCK_OUT_CHANNELS = (128, 256, 512, 512)   # shipped 884-16c-hy VAE


def vae_decoder_param_shapes(block_out_channels=VAE_BLOCK_OUT_CHANNELS, latent_channels: int = 16, out_channels: int = 3,
                             layers_per_block: int = 2) -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of AutoencoderKLCausal3D's decode side (vae/autoencoder_kl_causal_3d.py:98-115,
    vae/vae.py:158-226, vae/unet_causal_3d_blocks.py:320-347,579-616,822-851); checked against the imported reference
    by tools/make_golden_vae.py (load_state_dict strict)."""
    t: Dict[str, Tuple[int, ...]] = {}

    def conv(name, co, ci, k):
        t[name + ".weight"] = (co, ci, k, k, k)
        t[name + ".bias"] = (co,)

    def norm(name, c):
        t[name + ".weight"] = (c,)
        t[name + ".bias"] = (c,)

    def resnet(pre, ci, co):
        norm(pre + "norm1", ci)
        conv(pre + "conv1.conv", co, ci, 3)
        norm(pre + "norm2", co)
        conv(pre + "conv2.conv", co, co, 3)
        if ci != co:
            conv(pre + "conv_shortcut.conv", co, ci, 1)

    conv("post_quant_conv", latent_channels, latent_channels, 1)
    top = block_out_channels[-1]
    conv("decoder.conv_in.conv", top, latent_channels, 3)
    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        t[a + n + ".weight"] = (top, top)
        t[a + n + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.1.", top, top)
    rev = list(reversed(block_out_channels))
    prev = rev[0]
    nb = len(block_out_channels)
    for i, oc in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else oc, oc)
        spatial = i < 3
        temporal = (i >= nb - 1 - 2) and (i != nb - 1)
        if spatial or temporal:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", oc, oc, 3)
        prev = oc
    norm("decoder.conv_norm_out", block_out_channels[0])
    conv("decoder.conv_out.conv", out_channels, block_out_channels[0], 3)
    return t


def vae_encoder_param_shapes(block_out_channels=VAE_BLOCK_OUT_CHANNELS, latent_channels: int = 16, in_channels: int = 3,
                             layers_per_block: int = 2) -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of AutoencoderKLCausal3D's encode side (vae/vae.py:32-113: conv_in, 4 DownEncoderBlockCausal3D with
    `downsamplers.0.conv.conv` where the block downsamples, mid block, conv_norm_out, conv_out -> 2*latent; quant_conv
    autoencoder_kl_causal_3d.py:110-113); checked against the imported reference by tools/make_golden_vae_enc.py."""
    t: Dict[str, Tuple[int, ...]] = {}

    def conv(name, co, ci, k):
        t[name + ".weight"] = (co, ci, k, k, k)
        t[name + ".bias"] = (co,)

    def norm(name, c):
        t[name + ".weight"] = (c,)
        t[name + ".bias"] = (c,)

    def resnet(pre, ci, co):
        norm(pre + "norm1", ci)
        conv(pre + "conv1.conv", co, ci, 3)
        norm(pre + "norm2", co)
        conv(pre + "conv2.conv", co, co, 3)
        if ci != co:
            conv(pre + "conv_shortcut.conv", co, ci, 1)

    conv("encoder.conv_in.conv", block_out_channels[0], in_channels, 3)
    nb = len(block_out_channels)
    prev = block_out_channels[0]
    for i, oc in enumerate(block_out_channels):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", prev if j == 0 else oc, oc)
        spatial = i < 3
        temporal = (i >= nb - 1 - 2) and (i != nb - 1)
        if spatial or temporal:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv.conv", oc, oc, 3)
        prev = oc
    top = block_out_channels[-1]
    resnet("encoder.mid_block.resnets.0.", top, top)
    a = "encoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        t[a + n + ".weight"] = (top, top)
        t[a + n + ".bias"] = (top,)
    resnet("encoder.mid_block.resnets.1.", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out.conv", 2 * latent_channels, top, 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    return t


def synth_vae_state_dict(block_out_channels=VAE_BLOCK_OUT_CHANNELS, seed: int = 0, device="cpu", dtype=torch.float32,
                         encoder: bool = False):
    shapes = dict(vae_decoder_param_shapes(block_out_channels))
    if encoder:
        shapes.update(vae_encoder_param_shapes(block_out_channels))
    return {k: synth_param("vae." + k, shp, seed, device).to(dtype) for k, shp in shapes.items()}
