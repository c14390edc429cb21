"""hyvideo/utils/file_utils.py:47-70 `save_videos_grid`: video tensor [B,C,T,H,W] -> a file.  The reference writes mp4 through
imageio + ffmpeg; neither exists in this environment (and nothing may be installed), so the container is chosen by what is
importable: imageio if present (mp4, as the reference), else an animated GIF or WebP through Pillow, else the uint8 frames as
.npy.  Frame composition (grid over the batch, optional [-1,1] -> [0,1] rescale, clamp, uint8) follows the reference."""
from __future__ import annotations

import os

import numpy as np
import torch


def _grid(frame: torch.Tensor, n_rows: int, padding: int = 2) -> torch.Tensor:
    """torchvision.utils.make_grid(frame [B,C,H,W], nrow=n_rows) restated: images left to right, `nrow` per row, 2-px zero padding."""
    b, c, h, w = frame.shape
    if b == 1:
        return frame[0]
    xm = min(n_rows, b)
    ym = (b + xm - 1) // xm
    grid = torch.zeros(c, ym * (h + padding) + padding, xm * (w + padding) + padding, dtype=frame.dtype)
    for k in range(b):
        y, x = k // xm, k % xm
        grid[:, y * (h + padding) + padding:y * (h + padding) + padding + h, x * (w + padding) + padding:x * (w + padding) + padding + w] = frame[k]
    return grid


def frames_uint8(videos: torch.Tensor, rescale: bool = False, n_rows: int = 1):
    outs = []
    for x in videos.permute(2, 0, 1, 3, 4):                # t b c h w
        x = _grid(x.float().cpu(), n_rows).permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        outs.append((torch.clamp(x, 0, 1) * 255).numpy().astype(np.uint8))
    return outs


def save_videos_grid(videos: torch.Tensor, path: str, rescale: bool = False, n_rows: int = 1, fps: int = 24) -> str:
    """Returns the path actually written (the extension changes when no mp4 writer is available)."""
    outs = frames_uint8(videos, rescale, n_rows)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    try:
        import imageio
        imageio.mimsave(path, outs, fps=fps)
        return path
    except ImportError:
        pass
    stem = os.path.splitext(path)[0]
    try:
        from PIL import Image
        imgs = [Image.fromarray(f[..., 0] if f.shape[-1] == 1 else f) for f in outs]
        out = stem + ".gif"
        imgs[0].save(out, save_all=True, append_images=imgs[1:], duration=max(1, int(round(1000 / fps))), loop=0)
        return out
    except ImportError:
        out = stem + ".npy"
        np.save(out, np.stack(outs))
        return out
