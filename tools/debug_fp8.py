import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops, synthetic as syn
from oracle import dit_ref as R
DEV='cuda'; FP8=torch.float8_e4m3fn
def U(shape,key,scale=1.0): return syn.hashed_uniform(shape,key,19)*(scale*math.sqrt(3.0))
for (M,N,K) in ((300,520,512),(256,256,512),(300,520,640),(300,512,512)):
    a=U((M,K),f"f.a{K}",1.5).to(torch.bfloat16); w=U((N,K),f"f.w{K}"); b=U((N,),'f.b',0.1).to(torch.bfloat16)
    wscale=(w.abs().max()/448).to(torch.bfloat16); w8=(w/wscale.float()).clamp(-448,448).to(FP8)
    aq,asc=ops.quant_rows_fp8(a.to(DEV))
    P=R.Fp8MfmaPrec(); ref=P.linear(a.float(),P.fp8(w8,wscale),b.float())
    print(M,N,K,'ref nan',int(torch.isnan(ref).sum()), 'w8 nan', int(torch.isnan(w8.float()).sum()), 'aq nan', int(torch.isnan(aq.float()).sum()), 'asc', float(asc.min()), float(asc.max()))
    for rep in range(3):
        out=torch.full((M,N), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.gemm_fp8(aq,asc,w8.to(DEV),wscale.reshape(1).to(DEV),b.to(DEV),out=out)
        y=out.float().cpu(); nan=torch.isnan(y)
        d=(y-ref).abs(); d[nan]=0
        print('  rep',rep,'nan',int(nan.sum()),'first',nan.nonzero()[:6].tolist(),'maxdiff',float(d.max()), 'untouched(7.0)', int((y==7.0).sum()))
