import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
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
from aligner_amd import _lib
lib = _lib.load()
for k, v in [a.split('=') for a in sys.argv[1:]]:
    lib.aligner_debug_set_option(k.encode(), int(v))
for D in (16, 32, 64):
    print("D=%d: %.1f us" % (D, ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D))))
# the gradient (cotangent on gamma, the training case) on top of a search that kept log_alpha
w = torch.randn(B, Tx, Ty, generator=g).to(dev)
for D in (16, 32, 64):
    r = aligner_amd.boundary_search(lp, tx, ty, D, want_log_alpha=True)
    t_f = ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True))
    t_b = ev(lambda: aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, None, w))
    t_s = ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True, want_map=False))
    print("D=%d: search with log_alpha + gamma %.1f us (without the MAP sequence %.1f us), gradient %.1f us" % (D, t_f, t_s, t_b))
