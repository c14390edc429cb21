import sys, math, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops, synthetic as syn
from oracle import dit_ref as R
DEV='cuda'; FP8=torch.float8_e4m3fn; E=R.Prec(True)
def U(shape,key,scale=1.0): return syn.hashed_uniform(shape,key,19)*(scale*math.sqrt(3.0))
M,N,K=1000,768,1024
a=U((M,K),'a',1.5).to(torch.bfloat16); w=U((N,K),'w'); b=U((N,),'b',0.1).to(torch.bfloat16)
wscale=(w.abs().max()/448).to(torch.bfloat16); w8=(w/wscale.float()).clamp(-448,448).to(FP8)
aq,asc=ops.quant_rows_fp8(a.to(DEV))
gate,res=U((N,),'g',0.5).to(torch.bfloat16),U((M,N),'r').to(torch.bfloat16)
P=R.Fp8MfmaPrec(); ref=P.linear(a.float(),P.fp8(w8,wscale),b.float())
exp=R.gate_residual(res.float()[None],ref[None],gate.float()[None],E)[0]
y=ops.gemm_fp8(aq,asc,w8.to(DEV),wscale.reshape(1).to(DEV),b.to(DEV)).float().cpu()
print('plain max diff', float((y-ref).abs().max()))
for inplace in (False, True, False, True):
    r_dev=res.to(DEV).clone()
    out=r_dev if inplace else torch.empty_like(r_dev)
    got=ops.gemm_fp8(aq,asc,w8.to(DEV),wscale.reshape(1).to(DEV),b.to(DEV),out=out,gate=gate.to(DEV),res=r_dev).float().cpu()
    d=(got-exp).abs()
    bad=(d>0.02+exp.abs()*2**-7)
    print('inplace',inplace,'max diff',float(d.max()),'bad',int(bad.sum()), 'where', bad.nonzero()[:5].tolist())
    # recompute the expectation from the GPU's own y
    exp2=R.gate_residual(res.float()[None],y[None],gate.float()[None],E)[0]
    print('   vs formula on GPU y: max', float((got-exp2).abs().max()))
