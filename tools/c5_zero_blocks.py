"""C5 long-form search [8,500,4000] bf16 with the dense int32 path written inside the launch: HIP-event time per launch by the
number of zero workgroups (debug option maxpath_zero_blocks; 0 = every idle CU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import synth, _lib
dev = torch.device("cuda:0"); lib = _lib.load()
B, Tx, Ty = 8, 500, 4000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 5, bits=8, denom=8.0)).to(dev).to(torch.bfloat16)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
def ev(fn, it=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
ref = aligner_amd.align(v, tx, ty, path_dtype=torch.int32).path.clone()
for z in (0, 16, 24, 32, 48, 64, 96, 128, 0):
    lib.aligner_debug_set_option(b"maxpath_zero_blocks", z)
    r = aligner_amd.align(v, tx, ty, path_dtype=torch.int32)
    ok = bool(torch.equal(r.path, ref))
    print("zero workgroups %3s: %.1f us with the dense int32 path, %.1f us durations only, path equal %s" % (
        z or "all", ev(lambda: aligner_amd.align(v, tx, ty, path_dtype=torch.int32)), ev(lambda: aligner_amd.align(v, tx, ty, want_path=False)), ok))
lib.aligner_debug_set_option(b"maxpath_zero_blocks", 0)
