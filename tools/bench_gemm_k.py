"""GEMM time vs K at fixed M, N (bf16 and fp8 pipelined loops): slope = time per K-tile, intercept = fixed cost per output tile."""
import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
dev = 'cuda'; FP8 = torch.float8_e4m3fn
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 118800
for N in (3072, 9216):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    per_cu = tiles / 256.0
    for K in (768, 1536, 3072, 6144, 12288):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: ops.gemm(a, w, None, out=out))
        aq, asc = ops.quant_rows_fp8(a); w8 = (w * 50).to(FP8); ws = torch.ones(1, dtype=torch.bfloat16, device=dev)
        ms8 = timeit(lambda: ops.gemm_fp8(aq, asc, w8, ws, None, out=out))
        print(f"N={N} K={K}: bf16 {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.0f} TF, {ms*1e3/per_cu:.1f} us/tile | fp8 {ms8:.3f} ms = {2.0*M*N*K/ms8/1e9:.0f} TF, {ms8*1e3/per_cu:.1f} us/tile", flush=True)
        del a, w, out, aq, w8
