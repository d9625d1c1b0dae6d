import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
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
for (B, Tx, Ty) in ((64, 200, 1000), (8, 500, 4000)):
    lp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    print("[%d,%d,%d] forward-sum: plain %.1f us / with gradient %.1f us; CTC form %.1f us / with gradient %.1f us" % (
        B, Tx, Ty, ev(lambda: aligner_amd.forward_sum(lp, tx, ty, want_grad=False)), ev(lambda: aligner_amd.forward_sum(lp, tx, ty)),
        ev(lambda: aligner_amd.forward_sum(lp, tx, ty, want_grad=False, blank_logprob=-1.0)), ev(lambda: aligner_amd.forward_sum(lp, tx, ty, blank_logprob=-1.0))))
