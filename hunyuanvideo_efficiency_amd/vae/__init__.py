"""Mirror of hyvideo/vae/__init__.py:70-127 (load_vae) for the decode path."""
from __future__ import annotations

import torch

from .autoencoder_kl_causal_3d import AutoencoderKLCausal3D  # noqa: F401


def load_vae(vae_type: str = "884-16c-hy", vae_precision: str = "fp16", sample_size=None, vae_path=None, logger=None,
             device=None, state_dict=None):
    """Builds the 884-16c-hy topology (block_out_channels (128,256,512,512), 16 latent channels).
    `state_dict`: reference-format weights (keys decoder.*, post_quant_conv.*; encoder.* ignored).  Checkpoint FILES are
    loaded by the caller with torch.load(weights_only=True); this function never unpickles."""
    if vae_type != "884-16c-hy":
        raise NotImplementedError(f"VAE type {vae_type}: only the shipped 884-16c-hy decoder topology has kernels")
    dtype = {"fp16": torch.float16}.get(vae_precision)
    if dtype is None:
        raise NotImplementedError("VAE kernels are fp16 (the reference default --vae-precision fp16)")
    kw = {}
    if sample_size:
        kw["sample_size"] = sample_size
    vae = AutoencoderKLCausal3D(device=device, dtype=dtype, **kw)
    if state_dict is not None:
        vae.load_state_dict(state_dict)
    vae.requires_grad_(False)
    vae.eval()
    spatial_compression_ratio, time_compression_ratio = 8, 4
    return vae, vae_path, spatial_compression_ratio, time_compression_ratio
