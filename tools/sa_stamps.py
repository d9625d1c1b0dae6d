"""In-kernel timeline of softattn_kernel at C2: shader-clock stamps per wave (debug hook
aligner_debug_set_stamps), printed relative to the earliest workgroup entry."""
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
NW = int(os.environ.get("SA_WAVES", "8"))
nblk = 4096
st = torch.zeros((nblk, NW, 8), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.soft_attention(k, q, out=out)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
used = s[:, 0, 0] > 0
s = s[used]
names = ["entry", "staged+barrier", "mel frags ready", "mfma+softmax", "lse", "stores issued", "stores drained"]
print("workgroups stamped:", s.shape[0], " waves:", NW, " (cycles since the workgroup's first entry stamp; median / max over workgroups)")
t0 = s[:, :, 0].min(axis=1)                      # the shader clock differs per XCD: stay inside a workgroup
for w in range(NW):
    row = []
    for kk in (0, 1, 2, 3, 5, 6):
        v = s[:, w, kk]
        ok = v > 0
        if ok.any():
            d = (v - t0)[ok]
            row.append(f"{names[kk]} {np.median(d):6.0f}/{d.max():6.0f}")
    print(f"wave {w}: " + " | ".join(row))
last = (s[:, :, 1:7].max(axis=(1, 2)) - t0)
print("entry -> last stamp per workgroup: median", np.median(last), "max", last.max(), "cycles = us @2.4GHz:", last.max() / 2400.0)
rt0, rt1 = s[:, :, 7], s[:, :, 4]
print("100 MHz clock: first entry -> last entry %.2f us, first entry -> last drained %.2f us" %
      ((rt0.max() - rt0.min()) / 100.0, (rt1.max() - rt0.min()) / 100.0))
ent = np.sort((rt0.min(axis=1) - rt0.min()) / 100.0)
print("workgroup entry times (us), deciles:", np.round(ent[:: max(1, len(ent) // 10)], 2))
end = np.sort((rt1.max(axis=1) - rt0.min()) / 100.0)
print("workgroup drained times (us), deciles:", np.round(end[:: max(1, len(end) // 10)], 2))
