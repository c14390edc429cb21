import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops
DEV='cuda'; FP8=torch.float8_e4m3fn; BF=torch.bfloat16
def run(M,N,K,fp8):
    a=torch.ones(M,K); w=((torch.arange(N)%7)+1).float()[:,None].expand(N,K).contiguous()
    out=torch.full((M,N),-3.0,dtype=BF,device=DEV)
    if fp8:
        one=torch.ones(M,dtype=torch.float32,device=DEV); ws=torch.ones(1,dtype=BF,device=DEV)
        ops.gemm_fp8(a.to(FP8).to(DEV),one,w.to(FP8).to(DEV),ws,None,out=out)
    else:
        ops.gemm(a.to(BF).to(DEV),w.to(BF).to(DEV),None,out=out)
    y=out.float().cpu(); ref=(a@w.T).to(BF).float()
    bad=(y!=ref)|torch.isnan(y)
    return int(bad.sum()), sorted(set(bad.nonzero()[:,1].tolist()))[:12]
for (M,N,K) in ((256,8,384),(256,264,384),(256,256,384),(256,248,384),(256,128+8,512),(512,520,1024),(256,504,512)):
    print(M,N,K,'bf16',run(M,N,K,False),'fp8',run(M,N,K,True))
