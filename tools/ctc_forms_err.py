"""usage (GPU box): python tools/ctc_forms_err.py -- loss and gradient error of the two forms of the CTC kernels (systolic /
one sweeping wave) against torch.nn.functional.ctc_loss in float64, at deep lattices."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
from aligner_amd import _lib
from oracle import forward_sum_oracle as FS
lib=_lib.load(); dev=torch.device('cuda:0')
for seed,(Tx,Ty) in enumerate([(503,807),(503,1100),(400,900),(251,640)]):
    rng=np.random.default_rng(seed)
    x=torch.log_softmax(torch.from_numpy(rng.standard_normal((1,Tx,Ty)).astype(np.float32)),dim=1)
    tx=np.array([Tx]); ty=np.array([Ty])
    wl,wg=FS.ctc_forward_sum(x.numpy(),tx,ty,-1.0)
    z=np.concatenate([np.full((1,Ty),-1.0),x[0].numpy().astype(np.float64)],0)
    p=np.exp(z-np.log(np.exp(z).sum(0,keepdims=True)))[1:]
    occ=p-wg[0]
    for ow in (0,1):
        lib.aligner_debug_set_option(b"fwdsum_one_wave", ow)
        l,g=aligner_amd.forward_sum(x.to(dev),torch.from_numpy(tx),torch.from_numpy(ty),blank_logprob=-1.0)
        torch.cuda.synchronize()
        d=np.abs(g.cpu().numpy()[0].astype(np.float64)-wg[0])
        rel=(d/(5e-3*occ+2e-5)).max()
        print(Tx,Ty,'one_wave' if ow else 'systolic','loss err',abs(float(l[0])-wl[0]),'grad err max',d.max(),'fraction of the test tolerance',rel)
    lib.aligner_debug_set_option(b"fwdsum_one_wave", 0)
