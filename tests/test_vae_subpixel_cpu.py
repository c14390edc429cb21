"""CPU: the sub-pixel weight preparation (vae_ops.subpixel_weights) against the oracle's upsample + causal conv in fp64.

The kernel that consumes these weights only needs per-class tap offsets; here the class convs are evaluated with plain torch
indexing from exactly the tensors the kernel would get (w_sub, tap table), so the algebra - which taps are summed for which
output parity, the causal first frame, the replicate clamps, the class -> output scatter - is pinned without a GPU."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from hunyuanvideo_efficiency_amd.vae_ops import subpixel_weights  # noqa: E402
from oracle import vae_ref as R  # noqa: E402


def _emulate(x, w_sub, table, ntap, up_t):
    """x [C,T,H,W] fp64; returns [Co, T2, 2H, 2W] fp64 computed class by class as the kernel addresses it."""
    C, T, H, W = x.shape
    ncls, co, _ = w_sub.shape
    T2 = 2 * T - 1 if up_t else T
    out = torch.zeros(co, T2, 2 * H, 2 * W, dtype=torch.float64)
    wq = w_sub.to(torch.float64).reshape(ncls, co, ntap, C)
    for c in range(ncls):
        pt = (c >> 2) if up_t else 0
        ph, pw = (c >> 1) & 1, c & 1
        frames = T - 1 if (up_t and pt) else T
        for kt in range(frames):
            vt = kt + pt
            for kh in range(H):
                for kw in range(W):
                    acc = torch.zeros(co, dtype=torch.float64)
                    for j in range(ntap):
                        e = int(table[c, j])
                        ot, oh, ow = (e & 15) - 8, ((e >> 4) & 15) - 8, ((e >> 8) & 15) - 8
                        ti = max(vt + ot, 0)
                        hi = min(max(kh + oh, 0), H - 1)
                        wi = min(max(kw + ow, 0), W - 1)
                        acc += wq[c, :, j] @ x[:, ti, hi, wi]
                    t = 2 * kt + pt if up_t else kt
                    out[:, t, 2 * kh + ph, 2 * kw + pw] = acc
    return out


@pytest.mark.parametrize("factor,thw", [((2, 2, 2), (3, 3, 4)), ((1, 2, 2), (2, 4, 3)), ((2, 2, 2), (1, 2, 2))])
@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_subpixel_weights_reproduce_upsample_then_conv(factor, thw, mode):
    C, Co = 8, 6
    T, H, W = thw
    up_t = factor[0] == 2
    x = syn.hashed_uniform((1, C, T, H, W), "spc.x", 0).to(torch.float16).to(torch.float64)
    w = (syn.hashed_uniform((Co, C, 3, 3, 3), "spc.w", 0) / math.sqrt(27 * C)).to(torch.float16)
    ref = R.causal_conv3d(R.upsample_causal(x.float(), factor).double(), w.double(), torch.zeros(Co, dtype=torch.float64), R.Prec(False))[0]
    w_sub, table, ntap = subpixel_weights(w, up_t, mode)
    assert w_sub.dtype == torch.float16 and table.dtype == torch.int32
    assert w_sub.shape == ((8 if up_t else 4), Co, ntap * C)
    got = _emulate(x[0], w_sub, table, ntap, up_t)
    err = float((got - ref.double()).abs().max())
    scale = float(ref.abs().max())
    # exact: hi + lo reproduce every summed weight to 2^-22; fast: one fp16 rounding per summed weight
    assert err <= (2e-6 if mode == "exact" else 2e-3) * scale, (err, scale)
    if mode == "fast":
        assert err > 0        # the rounding is real: the two modes must not silently be the same thing


def test_subpixel_padding_and_class_order():
    w = (syn.hashed_uniform((5, 7, 3, 3, 3), "spc.w2", 0)).to(torch.float16)
    w_sub, table, ntap = subpixel_weights(w, True, "fast", cin_pad=64, cout_pad=8)
    assert w_sub.shape == (8, 8, 8 * 64) and ntap == 8
    ws = w_sub.reshape(8, 8, 8, 64)
    assert float(ws[:, 5:].abs().max()) == 0 and float(ws[:, :, :, 7:].abs().max()) == 0
    # class 0 = all-even parity: its first tap is the single corner weight w[..., 0, 0, 0] at offset (-1, -1, -1)
    assert torch.equal(ws[0, :5, 0, :7], w[:, :, 0, 0, 0]) and int(table[0, 0]) == (7 | 7 << 4 | 7 << 8)
    # class 7 = all-odd: last tap is the single corner w[..., 2, 2, 2] at offset (0, +1, +1)
    assert torch.equal(ws[7, :5, 7, :7], w[:, :, 2, 2, 2]) and int(table[7, 7]) == (8 | 9 << 4 | 9 << 8)
