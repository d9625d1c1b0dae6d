#!/usr/bin/env python3
"""Time the forward-sum objective (forward only / with gradient) at the C2 shape, both kernel forms."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import aligner_amd
    dev = torch.device("cuda:0")
    B, Tx, Ty = 64, 200, 1000
    if len(sys.argv) > 3:
        B, Tx, Ty = map(int, sys.argv[1:4])
    g = torch.Generator().manual_seed(0)
    logp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32, device=dev)
    ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    for want_grad in (False, True):
        for _ in range(5):
            out = aligner_amd.forward_sum(logp, tx, ty, want_grad=want_grad)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            out = aligner_amd.forward_sum(logp, tx, ty, want_grad=want_grad)
        e1.record()
        torch.cuda.synchronize()
        loss = out[0] if isinstance(out, tuple) else out
        print(f"{'with gradient' if want_grad else 'forward only '}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us   "
              f"loss[0] = {float(loss[0]):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "both":
        for env in ({}, {"ALIGNER_FWDSUM_ONE_WAVE": "1"}):
            print("one sweeping wave:" if env else "four sweeping waves:")
            subprocess.run([sys.executable, __file__], env={**os.environ, **env}, check=True)
    else:
        main()
