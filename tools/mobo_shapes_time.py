import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
dev = torch.device("cuda:0")
def ev(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
g = torch.Generator().manual_seed(0)
for (B, Tx, Ty, D) in ((64, 200, 1000, 16), (64, 200, 1000, 32), (256, 200, 1000, 16), (16, 400, 2000, 24), (1, 500, 4000, 32)):
    lp = (torch.randn(B, Tx, Ty, generator=g) * 2).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    w = torch.randn(B, Tx, Ty, generator=g).to(dev)
    r = aligner_amd.boundary_search(lp, tx, ty, D, want_log_alpha=True)
    print("[%d,%d,%d] D=%d: MAP %.0f us, log_alpha + gamma (no MAP) %.0f us, both %.0f us, gradient %.0f us" % (B, Tx, Ty, D,
        ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D)),
        ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True, want_map=False)),
        ev(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True)),
        ev(lambda: aligner_amd.boundary_search_backward(lp, tx, ty, D, r.log_alpha, None, w))))
