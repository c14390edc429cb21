"""Does the f16 MFMA keep subnormal inputs?  a = 2^-20 (subnormal in fp16) times w = 2^10 -> 2^-10 per product if kept, 0 if flushed."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import vae_ops as V
a = torch.full((256, 192), 2.0 ** -20, dtype=torch.float16, device='cuda')
w = torch.full((128, 192), 2.0 ** 10, dtype=torch.float16, device='cuda')
o = V.gemm_f16(a, w, out_f32=True)
print("a subnormal:", float(a[0, 0]), " out[0,0] =", float(o[0, 0]), " expected", 192 * 2.0 ** -10)
a2 = torch.full((256, 192), 2.0 ** 10, dtype=torch.float16, device='cuda')
w2 = torch.full((128, 192), 2.0 ** -20, dtype=torch.float16, device='cuda')
o = V.gemm_f16(a2, w2, out_f32=True)
print("w subnormal: out[0,0] =", float(o[0, 0]), " expected", 192 * 2.0 ** -10)
