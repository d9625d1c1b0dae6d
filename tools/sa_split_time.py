import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
from aligner_amd import _lib
lib=_lib.load(); dev=torch.device('cuda:0')
def ev(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
g=torch.Generator().manual_seed(0)
for (B,Tx,Ty) in ((8,500,4000),(4,500,4000),(16,400,2000),(2,300,1000)):
    k=torch.randn(B,80,Tx,generator=g).to(dev); q=torch.randn(B,80,Ty,generator=g).to(dev)
    tx=torch.full((B,),Tx,dtype=torch.int32,device=dev)
    out=[]
    for sp in (1,2,4):
        lib.aligner_debug_set_option(b"softattn_split", sp)
        if sp==1: lib.aligner_debug_set_option(b"softattn_no_pair", 1)
        out.append(ev(lambda: aligner_amd.soft_attention(k,q,t_x=tx,logp_dtype=torch.bfloat16)))
        lib.aligner_debug_set_option(b"softattn_no_pair", 0)
    lib.aligner_debug_set_option(b"softattn_split", 0)
    out.append(ev(lambda: aligner_amd.soft_attention(k,q,t_x=tx,logp_dtype=torch.bfloat16)))
    print(B,Tx,Ty,"one wave / two / four per strip, default: %.1f %.1f %.1f %.1f us" % tuple(out))
