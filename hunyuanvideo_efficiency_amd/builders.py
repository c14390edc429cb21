"""Random-init (deterministic hash) model construction for benchmarks, tests and sample_video.py without a checkpoint."""
from __future__ import annotations

import types

import torch

from . import synthetic as syn


def build_model(cfg: syn.DiTConfig, device, seed: int = 0):
    from .modules.models import HYVideoDiffusionTransformer
    args = types.SimpleNamespace(text_states_dim=cfg.text_states_dim, text_states_dim_2=cfg.text_states_dim_2)
    with torch.device("meta"):
        model = HYVideoDiffusionTransformer(
            args, in_channels=cfg.in_channels, out_channels=cfg.out_channels, hidden_size=cfg.hidden_size,
            heads_num=cfg.heads_num, mlp_width_ratio=cfg.mlp_width_ratio,
            mm_double_blocks_depth=cfg.mm_double_blocks_depth, mm_single_blocks_depth=cfg.mm_single_blocks_depth,
            rope_dim_list=cfg.rope_dim_list, guidance_embed=cfg.guidance_embed, dtype=torch.bfloat16)
    model.to_empty(device=device)
    shapes = syn.dit_param_shapes(cfg)
    sd = model.state_dict()
    assert set(sd) == set(shapes), set(sd) ^ set(shapes)
    with torch.no_grad():
        for k, p in sd.items():
            assert tuple(p.shape) == tuple(shapes[k]), (k, p.shape, shapes[k])
            p.copy_(syn.synth_param(k, shapes[k], seed, device).to(p.dtype))
    return model.eval()
