"""Attention outputs of the library currently in place on a few small shapes -> a .pt file (compare two libraries bit for bit)."""
import sys, torch
sys.path.insert(0, ".")
from hunyuanvideo_efficiency_amd import ops, synthetic as syn
outs = []
for (nq, nkv, H) in ((331, 331, 2), (324, 324, 2), (352, 331, 2), (21, 21, 2), (331, 331, 24), (300, 700, 1), (64, 449, 1), (64, 513, 1)):
    q = (syn.hashed_uniform((nq, H * 128), "c.q", 1, "cuda") * 2).to(torch.bfloat16)
    k = (syn.hashed_uniform((nkv, H * 128), "c.k", 2, "cuda") * 2).to(torch.bfloat16)
    v = (syn.hashed_uniform((nkv, H * 128), "c.v", 3, "cuda") * 2).to(torch.bfloat16)
    o = torch.empty(nq, H * 128, dtype=torch.bfloat16, device="cuda")
    ops.attn_fwd(q, k, v, o, H)
    outs.append(o.float().cpu())
torch.save(outs, sys.argv[1])
# distance of every output from a plain fp32 softmax reference on the GPU
errs = []
i = 0
for (nq, nkv, H) in ((331, 331, 2), (324, 324, 2), (352, 331, 2), (21, 21, 2), (331, 331, 24), (300, 700, 1), (64, 449, 1), (64, 513, 1)):
    q = (syn.hashed_uniform((nq, H * 128), "c.q", 1, "cuda") * 2).to(torch.bfloat16).float().reshape(nq, H, 128).transpose(0, 1)
    k = (syn.hashed_uniform((nkv, H * 128), "c.k", 2, "cuda") * 2).to(torch.bfloat16).float().reshape(nkv, H, 128).transpose(0, 1)
    v = (syn.hashed_uniform((nkv, H * 128), "c.v", 3, "cuda") * 2).to(torch.bfloat16).float().reshape(nkv, H, 128).transpose(0, 1)
    p = torch.softmax(q @ k.transpose(1, 2) * 128 ** -0.5, -1)
    ref = (p @ v).transpose(0, 1).reshape(nq, H * 128).cpu()
    errs.append(float((outs[i] - ref).abs().max()))
    i += 1
print("max |out - fp32 reference| per shape:", [round(e, 4) for e in errs])
