"""Diagnostic: the two big single-block GEMMs at the 720p shapes on random and on all-zero operands (same instruction stream): how
much of the distance to the MFMA roof is the power cap acting through the data (see profiles/r03/mfma_power_roof.txt)."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
M = 119056
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, K) in ((21504, 3072), (3072, 15360), (9216, 3072)):
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    for mode in ("randn", "zeros", "a_zero", "w_zero"):
        a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
        w = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device='cuda').to(torch.bfloat16)
        if mode in ("zeros", "a_zero"): a.zero_()
        if mode in ("zeros", "w_zero"): w.zero_()
        ms = timeit(lambda: ops.gemm(a, w, b, out=out))
        print(f"gemm M={M} N={N} K={K} {mode:7s}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
        del a, w
    del out
