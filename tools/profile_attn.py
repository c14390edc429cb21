"""One flash-attention launch at the bench's main-segment shape (n_q = n_kv = 118,811 tokens, 24 heads, fused QKV rows),
for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950)."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 118811
H, d = 24, 3072
qkv = torch.randn(S, 3 * d, device='cuda').to(torch.bfloat16)
cat = torch.empty(S, 5 * d, dtype=torch.bfloat16, device='cuda')
for _ in range(2):
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], cat[:, :d], H)
torch.cuda.synchronize()
