"""In-kernel phase totals of the forward-sum forward kernel (systolic form): per wave, cycles per phase spent in each part."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty = 64, 200, 1000
if len(sys.argv) > 3: B, Tx, Ty = map(int, sys.argv[1:4])
g = torch.Generator().manual_seed(0)
lp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
for _ in range(3): aligner_amd.forward_sum(lp, tx, ty, want_grad=False)
torch.cuda.synchronize()
st = torch.zeros((B * 16, 8), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.forward_sum(lp, tx, ty, want_grad=False)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64).reshape(B, 16, 8)
nw = 4 if Tx <= 252 else 8
print(f"[{B},{Tx},{Ty}]: cycles per phase (median over utterances)")
print("sweeper w: operands from LDS | the tile's frames | publish + lgkm | barrier (waits for the slowest wave)")
for w in range(nw):
    r = s[:, w, :]; n = np.maximum(r[:, 4], 1)
    print(f"  sweeper {w}: " + " | ".join(f"{np.median(r[:, i] / n):7.0f}" for i in range(4)) + f"   sum {np.median(r[:, :4].sum(1) / n):7.0f}")
print("stager w: wait for the tile's loads | LDS writes + out-tile reads + stores | issue loads | barrier")
for w in range(nw):
    r = s[:, nw + w, :]; n = np.maximum(r[:, 4], 1)
    print(f"  stager  {w}: " + " | ".join(f"{np.median(r[:, i] / n):7.0f}" for i in range(4)) + f"   sum {np.median(r[:, :4].sum(1) / n):7.0f}")
