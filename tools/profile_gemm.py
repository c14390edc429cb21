"""One big DiT GEMM (fc1: M=118800, N=12288, K=3072, bias+GELU) x3, for rocprofv3 PMC passes."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
M, N, K = 118800, 12288, 3072
a = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
b = torch.randn(N, device='cuda').to(torch.bfloat16); out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
for _ in range(3):
    ops.gemm(a, w, b, out=out, act=ops.ACT_GELU_TANH)
torch.cuda.synchronize()
