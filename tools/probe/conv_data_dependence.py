"""Diagnostic: the 128 -> 128 conv (conv128s_kernel) and the 256 -> 256 sub-pixel upsampler at the largest tile on random and on
all-zero activations / weights: schedule-bound or power-bound?  (profiles/r03/mfma_power_roof.txt)"""
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
T, H, W, C = 65, 256, 256, 128
M = T * H * W
b = torch.zeros(C, device=dev, dtype=torch.float16)
out = torch.empty(M, C, device=dev, dtype=torch.float16)
for mode in ("randn", "silu(randn)", "zeros", "x zero", "w zero"):
    x = torch.randn(M, C, device=dev, dtype=torch.float16)
    if mode == "silu(randn)": x = torch.nn.functional.silu(x)
    w = (torch.randn(C, 27, C, device=dev) * 0.02).to(torch.float16)
    if mode in ("zeros", "x zero"): x.zero_()
    if mode in ("zeros", "w zero"): w.zero_()
    ms = timeit(lambda: V.conv3d_causal(x, w, b, T, H, W, C, C, out=out))
    print(f"conv 128->128 @ {T}x{H}x{W} {mode:12s}: {ms:.3f} ms  {2.0 * M * C * C * 27 / ms / 1e9:.0f} TFLOP/s", flush=True)
    del x, w
