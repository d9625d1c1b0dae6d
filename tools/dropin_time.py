"""The drop-in call alone: maximum_path(value, mask) at [64,200,1000] -- GPU time, wall per synchronous call, host time per
asynchronous call; with the mask verification switched off (the multiply always) for comparison."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import synth, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, Tx, Ty = 64, 200, 1000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 2)).to(dev)
txa, tya = synth.synth_lengths(B, Tx, 500, 1000, 2)
txr, tyr = torch.from_numpy(txa).to(dev), torch.from_numpy(tya).to(dev)
mask = ((torch.arange(Tx, device=dev)[None, :, None] < txr[:, None, None]) & (torch.arange(Ty, device=dev)[None, None, :] < tyr[:, None, None])).to(v.dtype)
full = torch.ones_like(v)
def ev(fn, it=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for verify in (1, 0):
    lib.aligner_debug_set_option(b"maxpath_no_mask_verify", 0 if verify else 1)
    for name, m in (("full-length mask", full), ("ragged mask", mask)):
        t_gpu = ev(lambda: aligner_amd.maximum_path(v, m))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            aligner_amd.maximum_path(v, m); torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t0) / 50 * 1e6
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): aligner_amd.maximum_path(v, m)
        t_host = (time.perf_counter() - t0) / 200 * 1e6; torch.cuda.synchronize()
        print("maximum_path(value, mask) [64,200,1000], %s, %s: %.1f us of GPU time per call, %.1f us wall per synchronous call, %.1f us per call issued back to back" % (
            name, "mask verified on the device" if verify else "mask always multiplied in", t_gpu, t_wall, t_host))
lib.aligner_debug_set_option(b"maxpath_no_mask_verify", 0)
t = ev(lambda: aligner_amd.maximum_path(v, mask, mask_is_prefix=True))
print("   ... mask_is_prefix=True (no verification, no multiply): %.1f us of GPU time" % t)
