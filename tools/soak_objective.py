#!/usr/bin/env python3
"""Randomized soak of the forward-sum objective (loss + gradient) against the float64 oracle, across the
three kernel forms (four / eight sweeping waves, one wave).  Development aid.

Errors are reported in units of the unit test's tolerance (loss 5e-4 + 2e-7|loss|, gradient 1e-3|g| + 2e-5),
which is stated for log-probs of softmax scale ~3.  The soak also draws scale 8 (log-probs around -30, log Z
around -7000 on near-square shapes): fp32 log-domain rounding grows with the magnitude of the scores, and
both kernel forms then reach 1-1.7x that tolerance (posterior 1.001 where the oracle has 1.000).  A case
fails the soak above 3x -- 5x for the one-wave kernel (T_text > 504: ONE offset per frame for up to 1024 rows, where the
systolic kernels keep one per 63-row wave; near-square [643,663] at scale 8 measures 3.6x).

    python tools/soak_objective.py [cases] [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd  # noqa: E402
from oracle import forward_sum_oracle as FS  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    worst_l = worst_g = 0.0
    bad = 0
    for it in range(n):
        B = int(rng.integers(1, 4))
        Tx = int(rng.choice([rng.integers(1, 64), rng.integers(60, 260), rng.integers(250, 520), rng.integers(500, 700)]))
        Ty = int(rng.integers(Tx, Tx + 900))
        z = rng.standard_normal((B, Tx, Ty)) * rng.choice([0.5, 3.0, 8.0])
        lp = (z - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - z.max(1, keepdims=True)).astype(np.float32)
        ty = rng.integers(max(Tx // 2, 1), Ty + 1, size=B)
        tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
        tx[0], ty[0] = Tx, Ty
        wl, wg = FS.forward_sum(lp, tx, ty)
        loss, grad = aligner_amd.forward_sum(torch.from_numpy(lp).to(dev), torch.from_numpy(tx), torch.from_numpy(ty))
        loss, grad = loss.cpu().numpy().astype(np.float64), grad.cpu().numpy().astype(np.float64)
        el = np.abs(loss - wl).max() / (5e-4 + 2e-7 * np.abs(wl).max())
        eg = (np.abs(grad - wg) / (1e-3 * np.abs(wg) + 2e-5)).max()
        worst_l, worst_g = max(worst_l, el), max(worst_g, eg)
        lim = 5 if Tx > 504 else 3
        if el > lim or eg > lim:
            bad += 1
            print(f"OUT OF TOLERANCE ({lim}x) case {it}: B={B} Tx={Tx} Ty={Ty} loss x{el:.2f} grad x{eg:.2f}", flush=True)
    print(f"done: {n} cases, {bad} out of tolerance; worst loss error {worst_l:.2f}x, worst gradient error {worst_g:.2f}x of the test tolerance")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
