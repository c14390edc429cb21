"""Diagnostic (ab_libs/w4stamps.so built with -DHV_W4_STAMPS swapped in as the product library): one attention launch at the bench
shape, then the s_memtime accumulators of workgroup 300 / wave 1: cycles per steady-state iteration spent (a) from the iteration's
start to the barrier's vmcnt wait, (b) in that wait, (c) in s_barrier.  Results of this build are not checked."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops, _lib
S, H, d = 119056, 24, 3072
qkv = torch.randn(S, 3 * d, device='cuda').to(torch.bfloat16)
out = torch.empty(S, d, dtype=torch.bfloat16, device='cuda')
for _ in range(2):
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, H)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_uint * 8)()
rc = lib.hv_attn_w4_debug_read(buf)
vm, bar, pre, total, n = buf[0], buf[1], buf[2], buf[3], buf[4]
print(f"cumulative cycles from the iteration start to gap 16 / 32 / 48 / 56: {buf[5] / n:.0f} {buf[6] / n:.0f} {buf[7] / n:.0f} {pre / n:.0f}")
print(f"rc={rc} iterations={n}: per iteration: start->barrier-wait {pre / n:.0f}, vmcnt wait {vm / n:.0f}, s_barrier {bar / n:.0f}, whole loop {total / n:.0f} (s_memtime ticks = shader cycles)")
