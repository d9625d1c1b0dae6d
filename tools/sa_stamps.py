import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, C, Tx, Ty = 64, 80, 200, 1000
g = torch.Generator().manual_seed(0)
k = torch.randn(B, C, Tx, generator=g).to(dev); q = torch.randn(B, C, Ty, generator=g).to(dev)
out = torch.empty((B, Tx, Ty), device=dev)
for _ in range(3): aligner_amd.soft_attention(k, q, out=out)
torch.cuda.synchronize()
nblk = B * ((Ty + 255) // 256)
st = torch.zeros((nblk, 8, 8), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.soft_attention(k, q, out=out)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
names = ["entry", "mel frags loaded", "text staged+barrier", "mfma+logits", "lse", "stored"]
for kk in range(1, 6):
    d = s[:, 0, kk] - s[:, 0, kk - 1]
    print(f"{names[kk]:>22}: median {np.median(d):8.0f} cycles  max {d.max():8.0f}")
tot = s[:, 0, 5] - s[:, 0, 0]
print("total per WG median", np.median(tot), "max", tot.max(), " span all WGs", s[:, :, 5].max() - s[:, :, 0][s[:, :, 0] > 0].min())
