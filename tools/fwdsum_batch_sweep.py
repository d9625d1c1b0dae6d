import os, sys, torch
sys.path.insert(0, os.getcwd())
import aligner_amd
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
Tx, Ty = 200, 1000
for B in (96, 104, 112, 120, 128):
    g = torch.Generator().manual_seed(0)
    logp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    res = []
    for opt in (1, -1):
        lib.aligner_debug_set_option(b"fwdsum_serial", opt)
        for blank in (None, -1.0):
            for _ in range(5): aligner_amd.forward_sum(logp, tx, ty, blank_logprob=blank)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): aligner_amd.forward_sum(logp, tx, ty, blank_logprob=blank)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"B={B}: serial plain {res[0]:.1f} ctc {res[1]:.1f} | side by side (forced) plain {res[2]:.1f} ctc {res[3]:.1f}")
lib.aligner_debug_set_option(b"fwdsum_serial", 0)
