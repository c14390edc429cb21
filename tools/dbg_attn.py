import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
torch.manual_seed(0)
dev='cuda'
def run(q,k,v):
    n_q=q.shape[0]; out=torch.empty(n_q,128,dtype=torch.bfloat16,device=dev)
    ops.attn_fwd(q.to(dev),k.to(dev),v.to(dev),out,1); torch.cuda.synchronize(); return out.float().cpu()
def ref(q,k,v):
    s=(q.float()@k.float().T)/math.sqrt(128); p=s.softmax(-1); return p@v.float()
bf=torch.bfloat16
n=64
# A: q=0 -> mean of V; V[key][d] = d
q=torch.zeros(n,128,dtype=bf); k=torch.randn(n,128).to(bf); v=torch.arange(128.)[None].expand(n,128).contiguous().to(bf)
o=run(q,k,v); print("A (expect 0..127):", o[0,:8].tolist(), o[0,120:].tolist(), "maxerr", (o-ref(q,k,v)).abs().max().item())
# B: q=0, V[key][d]=key -> 31.5
v=torch.arange(float(n))[:,None].expand(n,128).contiguous().to(bf)
o=run(q,k,v); print("B (expect 31.5):", o[0,:4].tolist(), o[40,:4].tolist())
# C: one-hot attention: k[j]=e_j*big, q[i]=e_{pi(i)}*big -> out[i]=v[pi(i)]
big=30.0
k=torch.zeros(n,128); k[torch.arange(n),torch.arange(n)]=big
perm=(torch.arange(n)*7+3)%n
q=torch.zeros(n,128); q[torch.arange(n),perm]=big
v=torch.arange(float(n))[:,None].expand(n,128).contiguous()
o=run(q.to(bf),k.to(bf),v.to(bf)); print("C expect", perm[:16].tolist()); print("C got   ", o[:16,0].tolist()); print("C got d5", o[:16,5].tolist())
# D: one-hot with v[key][d]=d + 1000*key?? keep small: v[key][d]= (key*128+d)%251
v=((torch.arange(n)[:,None]*128+torch.arange(128)[None])%251).float()
o=run(q.to(bf),k.to(bf),v.to(bf)); r=ref(q.to(bf),k.to(bf),v.to(bf)); print("D maxerr", (o-r).abs().max().item()); 
bad=(o-r).abs()>1
print("D bad rows", bad.any(1).nonzero().flatten().tolist()[:20], "bad cols", bad.any(0).nonzero().flatten().tolist()[:40])
