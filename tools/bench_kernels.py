"""Kernel microbenchmarks at the 720p x 129f shapes (torch.cuda.Event timing on the launch stream)."""
import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
dev = 'cuda'
def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
which = sys.argv[1] if len(sys.argv) > 1 else "all"
S, H, d = 119056, 24, 3072
if which in ("all", "attn"):
    for s in (8192, 32768, S):
        qkv = (torch.randn(s, 3 * d, device=dev) ).to(torch.bfloat16)
        out = torch.empty(s, d, dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: ops.attn_fwd(qkv[:, :d], qkv[:, d:2*d], qkv[:, 2*d:], out, H), n=2 if s > 50000 else 5)
        fl = 4.0 * s * s * 128 * H
        print(f"attn S={s}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s", flush=True)
        del qkv, out
if which in ("all", "gemm"):
    M = 118800
    for (N, K) in ((9216, 3072), (3072, 3072), (12288, 3072), (3072, 12288), (21504, 3072), (3072, 15360)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: ops.gemm(a, w, b, out=out), n=5)
        print(f"gemm M={M} N={N} K={K}: {ms:.2f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
        del a, w, b, out
if which in ("all", "rowwise"):
    M = 118800
    x = torch.randn(M, d, device=dev).to(torch.bfloat16); sh = torch.randn(d, device=dev).to(torch.bfloat16); o = torch.empty_like(x)
    ms = timeit(lambda: ops.ln_modulate(x, sh, sh, out=o), n=10)
    print(f"ln_modulate: {ms:.3f} ms  {2*M*d*2/ms/1e6:.0f} GB/s")
    qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16); w = torch.ones(128, device=dev, dtype=torch.bfloat16)
    cs = torch.randn(M, 128, device=dev)
    ms = timeit(lambda: ops.qknorm_rope_(qkv, w, w, cs, cs, M, H, d), n=10)
    print(f"qknorm_rope: {ms:.3f} ms  {4*M*d*2/ms/1e6:.0f} GB/s (q,k read+write)")
