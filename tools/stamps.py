#!/usr/bin/env python3
"""Print the in-kernel phase breakdown of the forward kernel (debug stamps)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _lib, synth
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty = 64, 200, 1000
if len(sys.argv) > 3: B, Tx, Ty = map(int, sys.argv[1:4])
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 2)).to(dev)
DT = {"f32": (torch.float32, _lib.DT_F32), "bf16": (torch.bfloat16, _lib.DT_BF16), "f16": (torch.float16, _lib.DT_F16)}[os.environ.get("STAMP_DTYPE", "f32")]
v = v.to(DT[0])
tx = torch.full((B,), int(os.environ.get('STAMP_TX', Tx)), dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
tok = torch.empty((B, Ty), dtype=torch.int32, device=dev); dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, Tx, Ty) + 256, dtype=torch.uint8, device=dev)
st = torch.zeros((2 * B, 16, 16), dtype=torch.int64, device=dev)   # (2B: an utterance may run as two workgroups)
def run():
    _lib.check(lib.aligner_maxpath_forward(v.data_ptr(), DT[1], None, 0, tx.data_ptr(), ty.data_ptr(), tok.data_ptr(), dur.data_ptr(), ws.data_ptr(), ws.numel(), B, Tx, Ty, -1e9, int(os.environ.get('STAMP_FLAGS', '0')), torch.cuda.current_stream().cuda_stream))
for _ in range(5): run()
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(st.data_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
if (s[B:, 0, 5] != 0).any():
    print("two workgroups per utterance: the second one's stamps (it runs the backtrack)")
    first = s[:B]; s = s[B:]
    print("   first workgroup's own end (cycles since its entry, median):", int(np.median(first[:, 0, 4] - first[:, 0, 0])))
else:
    s = s[:B]
nw = int((s[0, :, 0] != 0).sum())
t0 = s[:, :nw, 0].min(axis=1, keepdims=True)
names = ["entry", "fwd_done", "win_loaded", "walk_done", "-", "end"]
clk = (s[:, 0, 5] - s[:, 0, 0]) / ((s[:, 0, 7] - s[:, 0, 6]) / 100e6) / 1e9
print("waves", nw, "shader clock GHz (median)", round(float(np.median(clk)), 3))
for k in (1, 2, 3, 5):
    d = s[:, 0, k] - s[:, 0, 0]
    print(f"{names[k]:>11}: wave0 cycles since entry  median {np.median(d):9.0f}  min {d.min():9.0f}  max {d.max():9.0f}   ({np.median(d)/np.median(clk)/1e3:6.2f} us)")
print("per-wave own stream done, before the closing barrier (b=0):", [int(s[0, w, 4] - s[0, 0, 0]) for w in range(nw)])
print("per-wave fwd_done (b=0):", [int(s[0, w, 1] - s[0, 0, 0]) for w in range(nw)])
start = s[:, 0, 0]
for w in range(nw):
    r = s[0, w]
    if r[8] > 0: print(f"wave {w}: tile-10 stamps: start {int(r[8] - s[0, 0, 0])}, then +", [int(r[k] - r[8]) for k in (9, 10, 11)],
                       "(compute: sweep done, stores done, barrier passed; loader: LDS written, loads issued, barrier passed)")
print("block start skew cycles: max-min", int(start.max() - start.min()))
