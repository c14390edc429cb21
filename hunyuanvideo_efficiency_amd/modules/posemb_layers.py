"""RoPE-3D frequency tables (hyvideo/modules/posemb_layers.py:191-310 as called from inference.py:450-495).
Built once per video with a handful of torch ops on whatever device the caller wants (the reference builds them on
the CPU); their APPLICATION is the fused hv_qknorm_rope_bf16 kernel."""
from __future__ import annotations

from typing import List, Sequence, Union

import torch


def get_nd_rotary_pos_embed(rope_dim_list: Sequence[int], start, *args, theta: float = 10000.0, use_real: bool = False,
                            theta_rescale_factor: Union[float, List[float]] = 1.0,
                            interpolation_factor: Union[float, List[float]] = 1.0, device=None):
    """Positions are the integer grid of `start` (sizes) in 'ij' order, flattened first-axis-major; per axis i
    cos/sin(pos * theta^(-2j/dim_i)) with every frequency repeated twice; axes concatenated -> ([S, D], [S, D])."""
    if len(args) != 0:
        raise NotImplementedError("only the sizes form get_nd_rotary_pos_embed(rope_dim_list, sizes, ...) is used")
    if not use_real:
        raise NotImplementedError("complex tables are not used by the kernels; pass use_real=True")
    sizes = [start] * len(rope_dim_list) if isinstance(start, int) else list(start)
    n = len(rope_dim_list)
    assert len(sizes) == n

    def per_axis(v):
        if isinstance(v, (int, float)):
            return [float(v)] * n
        return [float(v[0])] * n if len(v) == 1 else [float(x) for x in v]
    rescale, interp = per_axis(theta_rescale_factor), per_axis(interpolation_factor)
    axes = [torch.arange(s, dtype=torch.float32, device=device) for s in sizes]
    grids = torch.meshgrid(*axes, indexing="ij")
    cos_parts, sin_parts = [], []
    for dim, g, rs, ip in zip(rope_dim_list, grids, rescale, interp):
        th = theta * (rs ** (dim / (dim - 2))) if rs != 1.0 else theta
        inv = 1.0 / (th ** (torch.arange(0, dim, 2, dtype=torch.float32, device=device)[: dim // 2] / dim))
        ang = torch.outer(g.reshape(-1) * ip, inv)
        cos_parts.append(ang.cos().repeat_interleave(2, dim=1))
        sin_parts.append(ang.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos_parts, dim=1), torch.cat(sin_parts, dim=1)
