import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops, _lib
lib = _lib.load()
S = int(sys.argv[1]); v = int(sys.argv[2]); H, d = 24, 3072
lib.hv_debug_set_attn_variant(v)
qkv = torch.randn(S, 3 * d, device='cuda').to(torch.bfloat16)
out = torch.empty(S, d, dtype=torch.bfloat16, device='cuda')
for _ in range(2):
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2*d], qkv[:, 2*d:], out, H)
torch.cuda.synchronize()
