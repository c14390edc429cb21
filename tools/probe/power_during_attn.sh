#!/bin/bash
# Diagnostic: socket power / shader clock (rocm-smi) sampled while the 720p attention launch runs back to back for ~20 s.
# usage (on the GPU box): bash tools/probe/power_during_attn.sh [lib.so]
export HV_ALLOW_EXPERIMENT_LIB=1     # experiment libraries are swapped in below
if [ -n "$1" ]; then cp "$1" hunyuanvideo_efficiency_amd/lib/libhv_kernels.so; fi
python3 - <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
S, H, d = 119056, 24, 3072
qkv = torch.randn(S, 3 * d, device='cuda').to(torch.bfloat16)
out = torch.empty(S, d, dtype=torch.bfloat16, device='cuda')
t0 = time.time()
n = 0
while time.time() - t0 < 22:
    for _ in range(8):
        ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, H)
    torch.cuda.synchronize()
    n += 8
print(f"{n} launches, {(time.time() - t0) / n * 1e3:.2f} ms each")
PY
pid=$!
sleep 8
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|junction\|mclk" | tr '\n' ';'; echo
  sleep 2
done
wait $pid
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
