import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd._lib import LIB
dev='cuda'; hd=64
torch.manual_seed(0)
def run(B,H,S,causal):
    d=H*hd
    qkv=(torch.randn(B*S,3*d,device=dev)*0.5).to(torch.bfloat16)
    o=torch.empty(B*S,d,device=dev,dtype=torch.bfloat16); do=torch.randn(B*S,d,device=dev).to(torch.bfloat16)
    lse=torch.empty(B,H,S,device=dev); delta=torch.empty(B,H,S,device=dev)
    q=(qkv,0,3*d,S*3*d); k=(qkv,d,3*d,S*3*d); v=(qkv,2*d,3*d,S*3*d); oo=(o,0,d,S*d)
    sc=hd**-0.5
    ops.flash_fwd(q,k,v,oo,lse,None,B,H,S,S,hd,sc,causal)
    outs=[]
    for fn in (ops.flash_bwd, ops.flash_bwd1):
        dqkv=torch.full((B*S,3*d),float('nan'),device=dev,dtype=torch.bfloat16)
        dq=(dqkv,0,3*d,S*3*d); dk=(dqkv,d,3*d,S*3*d); dv=(dqkv,2*d,3*d,S*3*d)
        fn(q,k,v,oo,do,lse,None,dq,dk,dv,delta,B,H,S,S,hd,sc,causal)
        torch.cuda.synchronize(); outs.append(dqkv.float())
    a,b=outs
    for name,i in (('dq',0),('dk',1),('dv',2)):
        e=(b[:,i*d:(i+1)*d]-a[:,i*d:(i+1)*d]).abs()
        rows=e.max(dim=1).values; cols=e.max(dim=0).values
        bad=(rows>2e-3*max(1.0, float(a[:,i*d:(i+1)*d].abs().max())/0.1)).nonzero().flatten().tolist()
        print(name,'S',S,'causal',causal,'max',float(e.max()),'bad rows',len(bad), bad[:40], 'bad cols', (cols>2e-3).nonzero().flatten().tolist()[:70])
run(1,1,64,False); run(1,1,128,False); run(1,1,256,False); run(1,1,512,False); run(1,1,256,True)
