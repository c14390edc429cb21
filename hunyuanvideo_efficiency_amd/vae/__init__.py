"""Mirror of hyvideo/vae/__init__.py:70-127 (load_vae) for the decode path."""
from __future__ import annotations

import torch

import json

from .autoencoder_kl_causal_3d import AutoencoderKLCausal3D, DiagonalGaussianDistribution  # noqa: F401

# constructor arguments config.json may carry (autoencoder_kl_causal_3d.py:60-82 in the reference)
_CONFIG_KEYS = ("in_channels", "out_channels", "down_block_types", "up_block_types", "block_out_channels", "layers_per_block",
                "act_fn", "latent_channels", "norm_num_groups", "sample_size", "sample_tsize", "scaling_factor", "force_upcast",
                "spatial_compression_ratio", "time_compression_ratio", "mid_block_add_attention")


def load_t_ops_config(json_path: str) -> dict:
    """vae/__init__.py:66-68."""
    with open(json_path, "r") as f:
        return json.load(f)


def _apply_t_ops_config_to_vae(vae: AutoencoderKLCausal3D, t_ops_config: dict):
    """vae/__init__.py:15-63: inject the fork's temporal-op configuration into encoder/decoder blocks."""
    vae.apply_t_ops_config(t_ops_config)


def load_vae(vae_type: str = "884-16c-hy", vae_precision: str = "fp16", sample_size=None, vae_path=None, logger=None,
             device=None, t_ops_config_path: str = None, test: bool = False, state_dict=None, with_encoder=None):
    """Builds the 884-16c-hy topology (block_out_channels (128,256,512,512), 16 latent channels).
    `vae_path`: directory with the reference's `config.json` + `pytorch_model.pt` (read by checkpoint.read_vae_checkpoint with
    torch.load(weights_only=True); "state_dict" wrapper and "vae." prefix handled as vae/__init__.py:97-101).
    `state_dict`: already-loaded reference-format weights (keys decoder.*, post_quant_conv.*, and encoder.* / quant_conv.*).
    The encode half is built when the weights carry `encoder.*` keys (override with `with_encoder`).
    `t_ops_config_path` + `test=True`: apply the fork's t_ops_config.json (vae/__init__.py:120-125; infer.py:96-105)."""
    if vae_type != "884-16c-hy":
        raise NotImplementedError(f"VAE type {vae_type}: only the shipped 884-16c-hy decoder topology has kernels")
    dtype = {"fp16": torch.float16}.get(vae_precision)
    if dtype is None:
        raise NotImplementedError("VAE kernels are fp16 (the reference default --vae-precision fp16)")
    kw = {}
    if vae_path is not None and state_dict is None:
        from ..checkpoint import read_vae_checkpoint
        if logger is not None:
            logger.info(f"Loading 3D VAE model ({vae_type}) from: {vae_path}")
        state_dict, cfg = read_vae_checkpoint(vae_path)
        if cfg:
            kw.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in _CONFIG_KEYS})
    if sample_size:
        kw["sample_size"] = sample_size
    if with_encoder is None:
        with_encoder = state_dict is not None and any(k.startswith("encoder.") for k in state_dict)
    vae = AutoencoderKLCausal3D(device=device, dtype=dtype, with_encoder=with_encoder, **kw)
    if state_dict is not None:
        vae.load_state_dict(state_dict)
    vae.requires_grad_(False)
    vae.eval()
    if t_ops_config_path is not None and test:
        if logger is not None:
            logger.info("Applying T-pool/pad configs to the loaded VAE.")
        _apply_t_ops_config_to_vae(vae, load_t_ops_config(t_ops_config_path))
    return vae, vae_path, vae.config.spatial_compression_ratio, vae.config.time_compression_ratio
