#!/usr/bin/env python3
"""In-kernel phase totals of the boundary search gradient's chain kernel (split form), per position segment."""
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
w = torch.randn(B, Tx, Ty, generator=g).to(dev)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
if len(sys.argv) > 5: lib.aligner_debug_set_option(b'mobo_start_lag', int(sys.argv[5]))
if len(sys.argv) > 6: lib.aligner_debug_set_option(b'mobo_stamp_wave', int(sys.argv[6]))
r = aligner_amd.boundary_search(lp, tx, ty, D, want_log_alpha=True)
for _ in range(3): aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, None, w)
torch.cuda.synchronize()
st = torch.zeros((4096, 24), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, None, w)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
nblk = int((s[:, 0] != 0).sum()); S = nblk // B
print(f"[{B},{Tx},{Ty}] D={D}: {nblk} blocks, {S} segments per utterance (block 0 = the rightmost)")
t0 = s[:nblk, 0].min()
clk = (s[:nblk, 2] - s[:nblk, 0]) / np.maximum((s[:nblk, 12] - s[:nblk, 11]) / 100e6, 1e-9) / 1e9
print("shader clock GHz (median)", round(float(np.median(clk)), 3))
print("blk  entry  loop_start   end   rows first |  per-row cycles: operands  phase1+report  barrier(waits for the helper)  phase2  store | exp2-per-term rows")
for sg in list(range(0, S, max(1, S // 8))) + [S - 1]:
    r = s[sg]
    rows = max(r[9], 1)
    print(f"{sg:3d} {r[0]-t0:7.0f} {r[1]-t0:9.0f} {r[2]-t0:9.0f} {int(r[9]):5d} {int(r[10]):5d} | " + " ".join(f"{r[3+q]/rows:8.0f}" for q in range(5)) + f" | {int(r[14])} | LDS drain before the barrier {r[8]/rows:.0f}")
print("cycles per row each compute wave spends at the barrier (own arrival to release), some blocks:")
for sg in (0, 1, 4, 16):
    r = s[sg]
    print(sg, [int(x / max(r[9], 1)) for x in r[15:23] if x > 0])
rt0 = s[:nblk, 11].min()
print("global time (us, 100 MHz counter) of every block of utterance 0: entry, end, rows, us per row")
for sg in range(S):
    r = s[sg]
    print(f"  blk {sg:2d}: {(r[11]-rt0)/100:8.1f} {(r[12]-rt0)/100:8.1f}  rows {int(r[9]):4d} first {int(r[10]):4d}  {(r[12]-r[11])/100/max(r[9],1):.3f} us/row  exact rows {int(r[14])}")
tr = st.cpu().numpy().reshape(-1)[2048 * 24:2048 * 24 + 128].reshape(16, 4, 2).astype(np.int64)
print("block 4, 16 consecutive rows: each wave's arrival at the barrier relative to the row's release of the row before, and the release")
for rr in range(1, 16):
    base = tr[rr - 1, :, 1].max()
    print(rr, "arrive", [int(x - base) for x in tr[rr, :, 0]], "release", int(tr[rr, :, 1].max() - base))
print("kernel span (cycles, first entry to last end):", int(s[:nblk, 2].max() - t0), "=", round((s[:nblk, 2].max() - t0) / np.median(clk) / 1e3, 1), "us")
