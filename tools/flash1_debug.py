import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pianobart_amd import ops
from pianobart_amd._lib import LIB
dev='cuda'; hd=64
torch.manual_seed(0)
def run(B,H,S,cross):
    d=H*hd
    if cross:
        qb=(torch.randn(B*S,d,device=dev)*0.5).to(torch.bfloat16); kvb=(torch.randn(B*S,4*d,device=dev)*0.5).to(torch.bfloat16)
        q=(qb,0,d,S*d); k=(kvb,2*d,4*d,S*4*d); v=(kvb,3*d,4*d,S*4*d)
    else:
        qkv=(torch.randn(B*S,3*d,device=dev)*0.5).to(torch.bfloat16)
        q=(qkv,0,3*d,S*3*d); k=(qkv,d,3*d,S*3*d); v=(qkv,2*d,3*d,S*3*d)
    o=torch.empty(B*S,d,device=dev,dtype=torch.bfloat16); do=torch.randn(B*S,d,device=dev).to(torch.bfloat16)
    lse=torch.empty(B,H,S,device=dev); delta=torch.empty(B,H,S,device=dev)
    km=torch.ones(B,S,device=dev); km[0]=0; km[1,100:]=0
    kx=torch.empty(B,dtype=torch.int32,device=dev); ops.key_extent(km,kx)
    oo=(o,0,d,S*d); sc=hd**-0.5
    ops.flash_fwd(q,k,v,oo,lse,km,B,H,S,S,hd,sc,False,kmax=kx)
    outs=[]
    for fn in (ops.flash_bwd, ops.flash_bwd1):
        if cross:
            dqb=torch.full((B*S,d),float('nan'),device=dev,dtype=torch.bfloat16); dkvb=torch.full((B*S,4*d),0.0,device=dev,dtype=torch.bfloat16)
            dq=(dqb,0,d,S*d); dk=(dkvb,2*d,4*d,S*4*d); dv=(dkvb,3*d,4*d,S*4*d)
        else:
            dqkv=torch.full((B*S,3*d),float('nan'),device=dev,dtype=torch.bfloat16)
            dq=(dqkv,0,3*d,S*3*d); dk=(dqkv,d,3*d,S*3*d); dv=(dqkv,2*d,3*d,S*3*d)
        db=[torch.zeros(d,device=dev) for _ in range(3)]
        ws=torch.empty(int(LIB.query('pb_flash_bias_ws_floats',B,H,S,S,hd)),device=dev)
        fn(q,k,v,oo,do,lse,km,dq,dk,dv,delta,B,H,S,S,hd,sc,False,kmax=kx,dbias=db,dbias_ws=ws)
        torch.cuda.synchronize()
        outs.append(((dqb.float(), dkvb[:,2*d:3*d].float(), dkvb[:,3*d:].float()) if cross else (dqkv[:,:d].float(), dqkv[:,d:2*d].float(), dqkv[:,2*d:].float()), [x.clone() for x in db]))
    (a,da),(b,dbb)=outs
    for n,x,y in zip(('dq','dk','dv'),a,b):
        e=(x-y).abs(); rows=e.max(dim=1).values
        print('cross' if cross else 'self', n, 'max', float(e.max()), 'nan', int(torch.isnan(y).sum()), 'bad rows', (rows>3e-3).nonzero().flatten().tolist()[:20])
    for n,x,y in zip(('dbq','dbk','dbv'),da,dbb):
        print('   ', n, float((x-y).abs().max()), float(x.abs().max()))
run(4,4,256,False); run(4,4,256,True)
