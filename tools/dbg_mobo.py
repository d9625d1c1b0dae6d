import numpy as np, torch, sys
sys.path.insert(0,'.')
import aligner_amd
from oracle import mobo_oracle as M
B,Tx,Ty,D=2,64,257,32
rng=np.random.default_rng(B*100+Tx)
e=(rng.standard_normal((B,Tx,Ty))*2).astype(np.float32)
tx=np.array([Tx,Tx],np.int32); ty=np.array([Ty,Ty],np.int32)
r=aligner_amd.boundary_search(torch.from_numpy(e).cuda(), torch.from_numpy(tx), torch.from_numpy(ty), D, want_log_alpha=True)
torch.cuda.synchronize()
la=r.log_alpha.cpu().numpy().astype(np.float64)
want=M.boundary_search(e[0].astype(np.float64), D)
fin=np.isfinite(want["log_alpha"])
err=np.where(fin, np.abs(la[0]-want["log_alpha"]), 0)
bad=np.argwhere(err>1e-2)
print("nbad",len(bad)); print(bad[:20])
for (i,j) in bad[:6]:
    print(i,j,la[0,i,j],want["log_alpha"][i,j])
print("first bad row", bad[:,0].min() if len(bad) else None)
