#!/usr/bin/env python3
"""Randomized soak of the CTC form of the forward-sum loss (development aid, not a test): the systolic kernels against the
one-sweeping-wave kernels on random ragged batches (both on the GPU: loss and gradient must agree to the kernels' own
rounding), every fifth case also against torch.nn.functional.ctc_loss in float64 on the CPU.

    python tools/soak_ctc.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd  # noqa: E402
from aligner_amd import _lib  # noqa: E402
from oracle import forward_sum_oracle as FS  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    lib = _lib.load()
    bad, worst_l, worst_g, t0 = 0, 0.0, 0.0, time.time()
    for it in range(n):
        B = int(rng.integers(1, 7))
        Tx = int(rng.choice([1, 2, 7, 31, 62, 63, 64, 100, 125, 126, 127, 200, 250, 251, 252, 300, 400, 502, 503]))
        Tx = max(1, Tx + int(rng.integers(-1, 2)))
        Ty = int(rng.integers(Tx, max(Tx + 1, min(4 * Tx + 50, 1300))))
        x = torch.log_softmax(torch.from_numpy(rng.standard_normal((B, Tx, Ty)).astype(np.float32) * float(rng.choice([0.5, 1.0, 3.0]))), dim=1)
        ty = rng.integers(max(Tx // 2, 1), Ty + 1, size=B)
        tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
        tx[0], ty[0] = Tx, Ty
        if rng.random() < 0.15:
            tx[-1] = min(Tx, int(ty[-1]) + 1)                     # more tokens than frames: loss +inf, gradient 0
        blank = float(rng.choice([-1.0, -3.0, -8.0]))
        xd = x.to(dev)
        txd, tyd = torch.from_numpy(tx), torch.from_numpy(ty)
        l1, g1 = aligner_amd.forward_sum(xd, txd, tyd, blank_logprob=blank)
        lib.aligner_debug_set_option(b"fwdsum_one_wave", 1)
        try:
            l2, g2 = aligner_amd.forward_sum(xd, txd, tyd, blank_logprob=blank)
        finally:
            lib.aligner_debug_set_option(b"fwdsum_one_wave", 0)
        torch.cuda.synchronize()
        l1, l2, g1, g2 = l1.cpu().numpy().astype(np.float64), l2.cpu().numpy().astype(np.float64), g1.cpu().numpy(), g2.cpu().numpy()
        fin = np.isfinite(l2)
        ok = np.array_equal(np.isfinite(l1), fin)
        msg = "" if ok else "finite pattern of the loss"
        if ok and fin.any():
            dl = np.abs(l1[fin] - l2[fin]) / (5e-4 + 1e-6 * np.abs(l2[fin]))
            dg = np.abs(g1 - g2).max()
            worst_l, worst_g = max(worst_l, dl.max()), max(worst_g, dg / 1e-2)
            # (each form is held to 5e-3 * occupancy + 2e-5 against torch; against one another: twice that, absolute)
            if dl.max() > 2.0 or dg > 1e-2:
                ok, msg = False, f"forms differ: loss {dl.max():.2f} of the tolerance, gradient {dg:.2e}"
        if ok and it % 5 == 0:
            wl, wg = FS.ctc_forward_sum(x.numpy(), tx, ty, blank)
            for b in range(B):
                if np.isfinite(wl[b]) != fin[b]:
                    ok, msg = False, f"finite pattern against torch b={b}"
                    break
                if fin[b] and abs(l1[b] - wl[b]) > 5e-4 + 1e-6 * abs(wl[b]):
                    ok, msg = False, f"loss against torch b={b}: {l1[b]} vs {wl[b]}"
                    break
                K, T = int(tx[b]), int(ty[b])
                if fin[b] and np.abs(g1[b, :K, :T] - wg[b, :K, :T]).max() > 5e-3:
                    ok, msg = False, f"gradient against torch b={b}"
                    break
        if not ok:
            bad += 1
            print(f"CASE {it} FAILED B={B} Tx={Tx} Ty={Ty} blank={blank} tx={tx.tolist()} ty={ty.tolist()}: {msg}", flush=True)
        if it % 20 == 19:
            print(f"{it + 1} cases, {bad} failures, worst loss / gradient difference between the forms as a fraction of the tolerance "
                  f"{worst_l:.2f} / {worst_g:.2f}, {time.time() - t0:.0f}s", flush=True)
    print(f"done: {n} cases, {bad} failures", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
