"""The decoder's tail at the largest tile (65 x 256 x 256 x 128 channels): GroupNorm apply + SiLU + conv_out as
hv_groupnorm_apply_f16 + the narrow implicit-GEMM conv (before) and as hv_conv3d_cout4_f16 (planes + gather-sum)."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import vae_ops as V
dev = 'cuda'
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for T, H, W in ((65, 256, 256), (33, 256, 256), (65, 208, 256)):
    M, C = T * H * W, 128
    x = torch.randn(M, C, device=dev, dtype=torch.float16)
    gw = torch.ones(C, device=dev, dtype=torch.float16); gb = torch.zeros(C, device=dev, dtype=torch.float16)
    w = (torch.randn(3, C, 3, 3, 3, device=dev) * 0.02)
    b8 = torch.zeros(8, device=dev, dtype=torch.float16)
    w8 = torch.zeros(8, 27, C, device=dev, dtype=torch.float16); w8[:3] = w.permute(0, 2, 3, 4, 1).reshape(3, 27, C).to(torch.float16)
    wf = V.cout4_weight_fragments(w)
    aff = V.groupnorm_affine(x, gw, gb)
    y = torch.empty_like(x)
    t0 = timeit(lambda: V.groupnorm_apply(x, aff, True, out=y))
    t1 = timeit(lambda: V.conv3d_causal(y, w8, b8, T, H, W, C, 8))
    t2 = timeit(lambda: V.conv_cout4(x, aff, True, wf, b8, T, H, W, C, 3))
    a = V.conv3d_causal(V.groupnorm_apply(x, aff, True), w8, b8, T, H, W, C, 8); bnew = V.conv_cout4(x, aff, True, wf, b8, T, H, W, C, 3)
    print(f"{T}x{H}x{W}: gn_apply {t0:.3f} ms + narrow conv {t1:.3f} ms = {t0 + t1:.3f} ms | planes + gather {t2:.3f} ms | max |diff| {float((a[:, :3].float() - bnew[:, :3].float()).abs().max()):.2e}", flush=True)
    del x, y, a, bnew
