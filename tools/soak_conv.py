#!/usr/bin/env python3
"""Randomized soak of the encoder convolutions (csrc/convgemm.hip: wide GEMM form, narrow form, chained epilogues, the
fused trailing layers, the one-call stack) against the float64 torch convolution.  Development aid.

    python tools/soak_conv.py [cases] [seed]

Every case draws a stack of 1-4 layers (channels from the set the forms switch on, K in {1,3,5}, B, T free) and runs it
three ways: `encode` (one call of the C ABI), `conv1d` layer by layer, and each layer a second time on the same input
(bit-identical results expected: the kernels have no atomics).  Tolerance: split-bf16 products carry ~2^-16 relative
error per product; allowed max(1e-4, 3e-5 * max|y|) per layer output."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import softattn  # noqa: E402

CH = [1, 3, 8, 16, 24, 31, 32, 33, 48, 64, 80, 96, 127, 128, 129, 160, 192, 256, 300, 384, 512, 640, 1024]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    worst, bad = 0.0, 0

    def ri(lo, hi):
        return int(torch.randint(lo, hi, (1,), generator=g))

    for it in range(n):
        depth = ri(1, 5)
        big = ri(0, 4) == 0
        B = ri(1, 5) if big else ri(1, 40)
        T = ri(1, 1300) if big else ri(1, 420)
        chans = [CH[ri(0, len(CH))] for _ in range(depth + 1)]
        while B * T * max(chans) > 40_000_000:
            T = max(1, T // 2)
        stack, ks = [], []
        for l in range(depth):
            K = [1, 3, 5][ri(0, 3)] if l == 0 or ri(0, 3) == 0 else 1
            w = torch.randn(chans[l + 1], chans[l], K, generator=g) / (chans[l] * K) ** 0.5
            b = torch.randn(chans[l + 1], generator=g) * 0.1 if ri(0, 4) else None
            stack.append((w, b))
            ks.append(K)
        x = torch.randn(B, chans[0], T, generator=g)
        # float64 reference, layer by layer (ReLU between the layers, none after the last)
        ref, h = [], x.double()
        for l, (w, b) in enumerate(stack):
            h = F.conv1d(h, w.double(), None if b is None else b.double(), padding=ks[l] // 2)
            if l + 1 < depth:
                h = torch.relu(h)
            ref.append(h)
        xd = x.to(dev)
        sd = [(w.to(dev), None if b is None else b.to(dev)) for w, b in stack]
        got = softattn.encode(xd, sd).cpu().double()
        tol = max(1e-4, 3e-5 * ref[-1].abs().max().item()) * depth
        err = (got - ref[-1]).abs().max().item()
        again = softattn.encode(xd, sd).cpu().double()
        same = torch.equal(got, again)
        # layer by layer, each against the reference on the REFERENCE's input (isolates the layer)
        lerr = 0.0
        for l, (w, b) in enumerate(sd):
            xin = (xd if l == 0 else ref[l - 1].float().to(dev))
            y = softattn.conv1d(xin, w, b, relu=l + 1 < depth).cpu().double()
            e = (y - ref[l]).abs().max().item()
            lt = max(1e-4, 3e-5 * ref[l].abs().max().item())
            if not e < lt:
                print(f"   layer {l}: {chans[l]}->{chans[l + 1]} k{ks[l]} err {e:.3e} tol {lt:.1e}", flush=True)
            lerr = max(lerr, e / lt)
        worst = max(worst, err / tol, lerr)
        if not (err < tol and lerr < 1.0 and same):
            bad += 1
            print(f"FAIL case {it}: B={B} T={T} chans={chans} K={ks} err={err:.3e} tol={tol:.1e} layer_rel={lerr:.2f} "
                  f"repeatable={same}", flush=True)
        elif it % 10 == 0:
            print(f"case {it}: B={B} T={T} chans={chans} K={ks} err={err:.2e} (tol {tol:.1e})", flush=True)
    print(f"soak_conv: {n} cases, {bad} failures, worst error / tolerance {worst:.3f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
