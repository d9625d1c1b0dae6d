"""usage (GPU box): python tools/mobo_bwd_time.py [option=value ...] -- the boundary search's gradient at [8,500,4000]
(bf16 energies, cotangent on gamma), per max duration; options are aligner_debug_set_option names."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
from aligner_amd import _lib
dev = torch.device("cuda:0")
def ev(fn, it=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
g = torch.Generator().manual_seed(0)
B, Tx, Ty = 8, 500, 4000
lp = (torch.randn(B, Tx, Ty, generator=g) * 2).bfloat16().to(dev)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
w = torch.randn(B, Tx, Ty, generator=g).to(dev)
lib = _lib.load()
for k, v in [a.split('=') for a in sys.argv[1:]]:
    assert lib.aligner_debug_set_option(k.encode(), int(v)) == 0, k
out = []
for D in (16, 32, 64):
    r = aligner_amd.boundary_search(lp, tx, ty, D, want_log_alpha=True)
    out.append("D=%d %.1f (gamma) %.1f (log_alpha) us" % (
        D, ev(lambda: aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, None, w)),
        ev(lambda: aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, w, None))))
print(" ".join(sys.argv[1:]) or "default", "|", "; ".join(out))
