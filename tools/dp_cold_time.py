#!/usr/bin/env python3
"""Forward-kernel time at [64,200,1000] with the scores warm (one tensor over and over: L2 / Infinity Cache hits) and
cold (a rotation of tensors larger than the 256 MB Infinity Cache together: every launch reads HBM)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _lib, synth
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty = 64, 200, 1000
if len(sys.argv) > 3: B, Tx, Ty = map(int, sys.argv[1:4])
K = max(2, int(600e6 // (B * Tx * Ty * 4)))
vals = [torch.from_numpy(synth.synth_value(B, Tx, Ty, 2 + i)).to(dev) for i in range(K)]
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
tok = torch.empty((B, Ty), dtype=torch.int32, device=dev); dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, Tx, Ty) + 256, dtype=torch.uint8, device=dev)
def run(v):
    _lib.check(lib.aligner_maxpath_forward_f32(v.data_ptr(), None, 0, tx.data_ptr(), ty.data_ptr(), tok.data_ptr(), dur.data_ptr(), ws.data_ptr(), ws.numel(), B, Tx, Ty, -1e9, 0, torch.cuda.current_stream().cuda_stream))
def timed(n, pick):
    for i in range(K): run(vals[pick(i)])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): run(vals[pick(i)])
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f"[{B},{Tx},{Ty}] fp32, {K} tensors of {B*Tx*Ty*4/1e6:.0f} MB")
print("warm (same tensor):      %.2f us per launch" % timed(200, lambda i: 0))
print("cold (rotating tensors): %.2f us per launch" % timed(200, lambda i: i % K))
