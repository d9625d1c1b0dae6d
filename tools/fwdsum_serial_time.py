"""Forward-sum with gradient on batches past half the CU count (forward, then the gradient-making backward kernel):
the hand-issued gradient stager against the compiler-scheduled one (debug option fwdsum_no_grad_stager)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
def ev(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
g = torch.Generator().manual_seed(0)
for (B, Tx, Ty) in ((160, 200, 1000), (256, 200, 1000), (64, 200, 1000)):
    lp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32); ty = torch.full((B,), Ty, dtype=torch.int32)
    row = []
    for serial, nohand in ((1, 0), (1, 1), (0, 0)):
        lib.aligner_debug_set_option(b"fwdsum_serial", serial); lib.aligner_debug_set_option(b"fwdsum_no_grad_stager", nohand)
        row.append(ev(lambda: aligner_amd.forward_sum(lp, tx, ty)))
    lib.aligner_debug_set_option(b"fwdsum_serial", 0); lib.aligner_debug_set_option(b"fwdsum_no_grad_stager", 0)
    print("[%d,%d,%d] with gradient: forward then backward, hand-issued gradient stager %.1f us | compiler-scheduled stager %.1f us | the library's choice %.1f us" % (B, Tx, Ty, row[0], row[1], row[2]))
