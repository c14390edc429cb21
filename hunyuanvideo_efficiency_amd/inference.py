"""Sampler-side pieces of the hot path with the reference's names (hyvideo/inference.py): the sequence-parallel
monkey-patch `parallelize_transformer` (:40-104), the RoPE table builder `get_rotary_pos_embed` (:450-495) and the
distributed initialisation (:157-176).  One process per GPU (torchrun), RCCL through torch.distributed "nccl"."""
from __future__ import annotations

import functools
import os
from typing import Optional

import torch
import torch.distributed as dist

from .long_ctx_attention import UlyssesLongContextAttention
from .modules.posemb_layers import get_nd_rotary_pos_embed


def init_distributed(ulysses_degree: int = 1, ring_degree: int = 1, backend: str = "nccl"):
    """inference.py:157-176: WORLD_SIZE must equal ring*ulysses.  ring_degree > 1 builds the two process groups of the hybrid
    scheme (Ulysses groups = runs of `ulysses_degree` consecutive ranks, ring groups = ranks with equal position in their
    run - yunchang's `use_ulysses_low` order) and registers them as the defaults of UlyssesLongContextAttention."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == ulysses_degree * ring_degree, "number of GPUs should be equal to ring_size * ulysses_degree."
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=device if backend == "nccl" else None)
    if world > 1:
        set_sequence_parallel_groups(ulysses_degree, ring_degree)
    return device


def set_sequence_parallel_groups(ulysses_degree: int, ring_degree: int):
    """Every rank creates every group, in the same order (torch.distributed.new_group is collective)."""
    rank = dist.get_rank()
    if ring_degree == 1:
        UlyssesLongContextAttention.set_sequence_parallel_group(None, None)
        return None, None
    ulysses_pg = ring_pg = None
    for i in range(ring_degree):
        ranks = list(range(i * ulysses_degree, (i + 1) * ulysses_degree))
        g = dist.new_group(ranks)
        if rank in ranks:
            ulysses_pg = g
    for j in range(ulysses_degree):
        ranks = list(range(j, ulysses_degree * ring_degree, ulysses_degree))
        g = dist.new_group(ranks)
        if rank in ranks:
            ring_pg = g
    UlyssesLongContextAttention.set_sequence_parallel_group(ulysses_pg, ring_pg)
    return ulysses_pg, ring_pg


def parallelize_transformer_module(transformer, group=None, kernels=None, ulysses_group=None, ring_group=None):
    """Wrap transformer.forward as the reference's new_forward does (inference.py:45-104): pick the split axis
    (latent H if (H/2) % P == 0, else W), shard x and the RoPE tables, install the SP attention object on every
    block, run, all-gather the output along the split axis."""
    original_forward = transformer.forward
    # `group`: the ranks the token axis is sharded over (None = WORLD = ulysses x ring); the attention object trades tokens for
    # heads inside `ulysses_group` (default: registered default, else `group`) and rings K/V over `ring_group`
    if ulysses_group is None and ring_group is None and UlyssesLongContextAttention._default_ring_group is None:
        ulysses_group = group
    sp_attn = UlyssesLongContextAttention(ulysses_group, kernels, ring_group)

    @functools.wraps(original_forward)
    def new_forward(x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None, freqs_sin=None,
                    guidance=None, return_dict=True):
        P, rank = dist.get_world_size(group), dist.get_rank(group)
        if x.shape[-2] // 2 % P == 0:
            split_dim = -2
        elif x.shape[-1] // 2 % P == 0:
            split_dim = -1
        else:
            raise ValueError(f"Cannot split video sequence into ulysses_degree x ring_degree ({P}) parts evenly")
        temporal_size, h, w = x.shape[2], x.shape[3] // 2, x.shape[4] // 2
        x = torch.chunk(x, P, dim=split_dim)[rank].contiguous()

        def shard(f):
            dim_thw = f.shape[-1]
            f = f.reshape(temporal_size, h, w, dim_thw)
            return torch.chunk(f, P, dim=split_dim - 1)[rank].reshape(-1, dim_thw).contiguous()
        freqs_cos, freqs_sin = shard(freqs_cos), shard(freqs_sin)
        for block in list(transformer.double_blocks) + list(transformer.single_blocks):
            block.hybrid_seq_parallel_attn = sp_attn
        output = original_forward(x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, return_dict)
        sample = output["x"] if isinstance(output, dict) else output
        parts = [torch.empty_like(sample) for _ in range(P)]
        dist.all_gather(parts, sample.contiguous(), group=group)
        sample = torch.cat(parts, dim=split_dim)
        if isinstance(output, dict):
            output["x"] = sample
            return output
        return sample

    transformer.forward = new_forward
    return transformer


def parallelize_transformer(pipe):
    """Reference entry point (inference.py:40): patches pipe.transformer in place."""
    parallelize_transformer_module(pipe.transformer)
    vae = getattr(pipe, "vae", None)
    if vae is not None and hasattr(vae, "enable_tile_parallel"):
        vae.enable_tile_parallel()      # beyond the reference (which decodes every tile on every rank): SURVEY.md 8e


def get_rotary_pos_embed(transformer, video_length: int, height: int, width: int, vae: str = "884-16c-hy",
                         rope_theta: float = 256.0, device=None):
    """inference.py:450-495 with `self.model` / `self.args` made explicit."""
    if "884" in vae:
        latents_size = [(video_length - 1) // 4 + 1, height // 8, width // 8]
    elif "888" in vae:
        latents_size = [(video_length - 1) // 8 + 1, height // 8, width // 8]
    else:
        latents_size = [video_length, height // 8, width // 8]
    ps = transformer.patch_size
    ps = [ps] * 3 if isinstance(ps, int) else list(ps)
    assert all(s % p == 0 for s, p in zip(latents_size, ps)), \
        f"Latent size(last 3 dimensions) should be divisible by patch size({ps}), but got {latents_size}."
    rope_sizes = [s // p for s, p in zip(latents_size, ps)]
    head_dim = transformer.hidden_size // transformer.heads_num
    rope_dim_list = transformer.rope_dim_list or [head_dim // 3] * 3
    assert sum(rope_dim_list) == head_dim, "sum(rope_dim_list) should equal to head_dim of attention layer"
    return get_nd_rotary_pos_embed(rope_dim_list, rope_sizes, theta=rope_theta, use_real=True, theta_rescale_factor=1,
                                   device=device)
