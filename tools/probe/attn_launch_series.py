"""Diagnostic (data dependence of the power-limited clock: argument 2 = randn | zeros | const | small): per-launch durations of N back-to-back 720p attention launches (events between launches), then the same with a host
synchronisation after every launch.  usage: python tools/probe/attn_launch_series.py [N]"""
import sys, time, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S, H, d = 119056, 24, 3072
mode = sys.argv[2] if len(sys.argv) > 2 else "randn"
qkv = torch.randn(S, 3 * d, device='cuda').to(torch.bfloat16)
if mode == "zeros":
    qkv.zero_()
elif mode == "const":
    qkv.fill_(0.5)
elif mode == "small":
    qkv.mul_(0.05)
elif mode == "vzero":
    qkv[:, 2 * d:].zero_()
elif mode == "qkzero":
    qkv[:, :2 * d].zero_()
print("inputs:", mode)
out = torch.empty(S, d, dtype=torch.bfloat16, device='cuda')
run = lambda: ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, H)
run(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.time()
ev[0].record()
for i in range(N):
    run(); ev[i + 1].record()
torch.cuda.synchronize()
wall = time.time() - t0
print("back to back :", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.1f}" for i in range(N)), f"| wall {wall / N * 1e3:.1f} ms/launch")
ts = []
for i in range(8):
    t0 = time.time(); run(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
print("sync each    :", " ".join(f"{t:.1f}" for t in ts))
