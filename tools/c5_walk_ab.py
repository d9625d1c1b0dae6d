"""C5 long-form alignment search [8,500,4000], bf16 scores: each half walking its own rows (round 4) against one walker for
all rows (debug option maxpath_no_split_walk), HIP events; with stamps the in-kernel timeline (cycles since entry)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import synth, _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty = 8, 500, 4000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 5, bits=8, denom=8.0)).to(dev).to(torch.bfloat16)
g = torch.Generator().manual_seed(0)
lp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g) * 0.1, dim=1).to(dev).to(torch.bfloat16)   # near-uniform log-probs: long tokens
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
def ev(fn, it=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for name, x in (("Appendix-A scores", v), ("near-uniform log-probs", lp)):
    for rnd in range(2):
        row = []
        for no_split in (0, 1):
            lib.aligner_debug_set_option(b"maxpath_no_split_walk", no_split)
            row.append((ev(lambda: aligner_amd.align(x, tx, ty, want_path=False)), ev(lambda: aligner_amd.align(x, tx, ty, path_dtype=torch.int32))))
        lib.aligner_debug_set_option(b"maxpath_no_split_walk", 0)
        print(f"{name}: durations only / with int32 dense path -- each half its own rows {row[0][0]:.1f} / {row[0][1]:.1f} us, one walker {row[1][0]:.1f} / {row[1][1]:.1f} us", flush=True)
a = aligner_amd.align(v, tx, ty, path_dtype=torch.int32)
lib.aligner_debug_set_option(b"maxpath_no_split_walk", 1)
b = aligner_amd.align(v, tx, ty, path_dtype=torch.int32)
lib.aligner_debug_set_option(b"maxpath_no_split_walk", 0)
print("same path:", bool((a.path == b.path).all()), "same durations:", bool((a.durations == b.durations).all()), "status", aligner_amd.read_status(dev))
