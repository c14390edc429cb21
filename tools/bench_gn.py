"""GroupNorm kernels at the VAE's largest activation (65x256x256 x 128 ch) next to torch elementwise kernels of the same traffic."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import vae_ops as V
dev = 'cuda'
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for M, C in ((65 * 256 * 256, 128), (33 * 256 * 256, 256), (17 * 128 * 128, 512)):
    x = torch.randn(M, C, device=dev, dtype=torch.float16)
    y = torch.empty_like(x)
    w = torch.ones(C, device=dev, dtype=torch.float16); b = torch.zeros(C, device=dev, dtype=torch.float16)
    gb = M * C * 2 / 1e9
    t = timeit(lambda: torch.mul(x, 2.0, out=y)); print(f"M={M} C={C}: torch mul      {t:.3f} ms  {2*gb/t:.0f} GB/s (r+w)")
    t = timeit(lambda: y.copy_(x)); print(f"               torch copy     {t:.3f} ms  {2*gb/t:.0f} GB/s (r+w)")
    t = timeit(lambda: torch.nn.functional.silu(x)); print(f"               torch silu     {t:.3f} ms  {2*gb/t:.0f} GB/s (r+w)")
    t = timeit(lambda: x.float().sum()); print(f"               torch f32 sum  {t:.3f} ms")
    aff = V.groupnorm_affine(x, w, b)
    t = timeit(lambda: V.groupnorm_affine(x, w, b)); print(f"               gn_affine      {t:.3f} ms  {gb/t:.0f} GB/s (r)")
    t = timeit(lambda: V.groupnorm_apply(x, aff, True, out=y)); print(f"               gn_apply silu  {t:.3f} ms  {2*gb/t:.0f} GB/s (r+w)")
    t = timeit(lambda: V.groupnorm_apply(x, aff, False, out=y)); print(f"               gn_apply plain {t:.3f} ms  {2*gb/t:.0f} GB/s (r+w)")
