#!/usr/bin/env python3
"""In-kernel phase totals of the boundary search's chain kernel (one position per thread form), per position segment."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty, D = 8, 500, 4000, 32
if len(sys.argv) > 4: B, Tx, Ty, D = map(int, sys.argv[1:5])
g = torch.Generator().manual_seed(0)
lp = (torch.randn(B, Tx, Ty, generator=g) * 2).bfloat16().to(dev)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
if len(sys.argv) > 5: lib.aligner_debug_set_option(b'mobo_start_lag', int(sys.argv[5]))
for _ in range(3): aligner_amd.boundary_search(lp, tx, ty, D)
torch.cuda.synchronize()
st = torch.zeros((4096, 24), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.boundary_search(lp, tx, ty, D)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
nblk = int((s[:, 0] != 0).sum()); S = nblk // B
print(f"[{B},{Tx},{Ty}] D={D}: {nblk} blocks, {S} segments per utterance")
t0 = s[:nblk, 0].min()
clk = (s[:nblk, 2] - s[:nblk, 0]) / np.maximum((s[:nblk, 12] - s[:nblk, 11]) / 100e6, 1e-9) / 1e9
print("shader clock GHz (median)", round(float(np.median(clk)), 3))
names = ["operands ready", "phase 1 + publish", "halo", "(to barrier)", "barrier", "phase 2", "stores"]
print("seg  entry  loop_start   end   rows first |  per-row cycles: operands  phase1+report  -  barrier(waits for the helper)  phase2  stores")
for sg in list(range(0, S, max(1, S // 8))) + [S - 1]:
    r = s[sg]   # utterance 0
    rows = max(r[9], 1)
    print(f"{sg:3d} {r[0]-t0:7.0f} {r[1]-t0:9.0f} {r[2]-t0:9.0f} {int(r[9]):5d} {int(r[10]):5d} | " + " ".join(f"{r[3+q]/rows:8.0f}" for q in range(6)) + f"   polls/row {r[13]/rows:.2f} exact-sum rows {int(r[14])}")
print("kernel span (cycles, first entry to last end):", int(s[:nblk, 2].max() - t0), "=", round((s[:nblk, 2].max() - t0) / np.median(clk) / 1e3, 1), "us")

hw = st.cpu().numpy()[:nblk, 18:24]
def dec(x):
    x = int(x); h = x & 0xffffffff
    return f"xcc{(x >> 32) & 15} se{(h >> 13) & 7} sh{(h >> 12) & 1} cu{(h >> 8) & 15} simd{(h >> 4) & 3} w{h & 15}"
print("placement of the waves of blocks 0..5:")
for blk in range(6):
    print(blk, [dec(x) for x in hw[blk] if x != 0])
cus = {}
for blk in range(nblk):
    x = int(hw[blk][0]); h = x & 0xffffffff
    key = ((x >> 32) & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)
    cus[key] = cus.get(key, 0) + 1
print("distinct CUs used:", len(cus), "max blocks on one CU:", max(cus.values()))
