import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rga3-release_amd")
from rga3.hip import ops
from oracle import kernels_ref as R
dev="cuda"
def rnd(shape, seed):
    g=torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)
for (S,Hq,Hkv) in ((2112,28,4),(2112,7,1),(2112,4,4),(2048,4,2),(2176,2,2)):
    D=128
    q=rnd((S,Hq,D),21); kv=rnd((S,2,Hkv,D),22); k,v=kv[:,0],kv[:,1]
    cu=torch.tensor([0,S],dtype=torch.int32,device=dev)
    out,lse=ops.attn_varlen(q,k,v,cu,cu,S,D**-0.5,True,return_lse=True)
    old,lse_old=ops.attn_varlen(q,k,v,cu,cu,S,D**-0.5,True,return_lse=True,impl=4)
    d=(out.float()-old.float()).norm(dim=-1)/(old.float().norm(dim=-1)+1e-9)   # [S,Hq]
    bad=(d>3e-2).nonzero()
    print(S,Hq,Hkv,'rel',float((out.float()-old.float()).norm()/old.float().norm()),'lse diff',float((lse-lse_old).abs().max()),'bad rows',bad.shape[0])
    if bad.shape[0]:
        rows=sorted(set(bad[:,0].tolist())); heads=sorted(set(bad[:,1].tolist()))
        print(' rows', rows[:10],'...',rows[-5:], 'n',len(rows),' heads',heads)
    again,_=ops.attn_varlen(q,k,v,cu,cu,S,D**-0.5,True,return_lse=True)
    print('  deterministic', torch.equal(out,again))
