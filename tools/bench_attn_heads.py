"""Attention kernel at the per-rank shape of Ulysses-P: n_q = n_kv = 118,811 tokens, 24/P heads; with / without KV split."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
S = 118811
for P in (8, 4, 2, 1):
    hp = 24 // P; w = hp * 128
    q, k, v = (torch.randn(S, w, device='cuda').to(torch.bfloat16) for _ in range(3))
    res = {}
    for split in (False, True):
        out = torch.empty(S, w, dtype=torch.bfloat16, device='cuda')
        ops.attn_fwd(q, k, v, out, hp, kv_split_workspace=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.attn_fwd(q, k, v, out, hp, kv_split_workspace=split)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        res[split] = (ms, out.float())
    fl = 4.0 * S * S * 128 * hp
    d = (res[True][1] - res[False][1]).abs().max().item()
    print(f"P={P} heads={hp}: single-pass {res[False][0]:.2f} ms ({fl/res[False][0]/1e9:.0f} TF/s)  with-split-workspace {res[True][0]:.2f} ms ({fl/res[True][0]/1e9:.0f} TF/s)  maxdiff {d:.2e}", flush=True)
    del q, k, v
