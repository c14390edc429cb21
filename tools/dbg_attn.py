import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
torch.manual_seed(0)
for (nq, nkv, H) in ((64, 64, 1), (64, 128, 1), (256, 64, 1), (300, 200, 2)):
    q = torch.randn(nq, H*128, device='cuda').to(torch.bfloat16); k = torch.randn(nkv, H*128, device='cuda').to(torch.bfloat16); v = torch.randn(nkv, H*128, device='cuda').to(torch.bfloat16)
    o = torch.empty_like(q)
    ops.attn_fwd(q, k, v, o, H)
    qf, kf, vf = (t.float().view(t.shape[0], H, 128).transpose(0, 1) for t in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(1, 2) / 128 ** 0.5, -1) @ vf
    ref = ref.transpose(0, 1).reshape(nq, H*128)
    ratio = (o.float() / ref)
    err = (o.float() - ref).abs().max().item()
    rr = ratio[:, :8].median(dim=1).values
    print(nq, nkv, H, "max err", err, "row ratio (first 8 rows)", rr[:8].tolist(), "rows 32..36", rr[32:36].tolist())
