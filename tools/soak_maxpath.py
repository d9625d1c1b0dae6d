#!/usr/bin/env python3
"""Randomized parity soak of the alignment search against the CPU oracle (development aid, not a test:
minutes of GPU time).  Shapes from tiny to long-form, ragged lengths, ties, non-finite scores.

    python tools/soak_maxpath.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd  # noqa: E402
from oracle import maxpath_oracle as O  # noqa: E402


def oracle(v, tx, ty, neg):
    p = np.zeros(v.shape, np.int32)
    O.maximum_path_c(p, v.copy(), tx, ty, neg)
    return p


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    bad = 0
    for it in range(n):
        kind = rng.integers(0, 6)
        B = int(rng.integers(1, 5))
        if kind == 5:                                   # long-form: several backtrack windows
            Tx = int(rng.integers(1, 520)); Ty = int(rng.integers(max(Tx, 2049), 5000))
        else:
            Tx = int(rng.integers(1, 530)); Ty = int(rng.integers(Tx, Tx + 1500))
        if rng.random() < 0.5:
            Ty = (Ty + 3) // 4 * 4                       # the 16-byte loader path
        v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        if kind == 1:
            v = rng.integers(-2, 3, (B, Tx, Ty)).astype(np.float32)
        elif kind == 2:
            v = (-rng.random((B, Tx, Ty)) * 30).astype(np.float32)
        elif kind == 3:
            v[rng.random(v.shape) < 0.01] = -np.inf
            v[rng.random(v.shape) < 0.003] = np.nan
            v[rng.random(v.shape) < 0.003] = np.inf
        elif kind == 4:
            v *= 1e8                                     # running scores far below max_neg_val
        ty = rng.integers(1, Ty + 1, B).astype(np.int32)
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
        if rng.random() < 0.5:
            tx[0], ty[0] = Tx, Ty
        neg = float(rng.choice([-1e9, -1e4, -3.0e38]))
        want = oracle(v, tx, ty, neg)
        vd = torch.from_numpy(v).to(dev)
        txd, tyd = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
        pdt = [torch.int32, torch.float32, torch.uint8, torch.bfloat16, torch.int64][int(rng.integers(0, 5))]
        # auto: the dense path inside the search launch (zero workgroups) where it applies; separate: the expand kernel;
        # prezeroed: the caller's zeros + the search kernel's ones
        for name, kw in (("auto", {}), ("generic", {"force_generic": True}), ("separate_expand", {"_test_flags": 4096}),
                         ("prezeroed", {"out_path_is_zero": True})):
            if name != "auto" and (it + len(name)) % 4:
                continue
            if name == "prezeroed":
                kw = dict(kw, out_path=torch.zeros((B, Tx, Ty), dtype=pdt, device=dev))
            got = aligner_amd.align(vd, txd, tyd, path_dtype=pdt, max_neg_val=neg, **kw).path.to(torch.int32).cpu().numpy()
            if not np.array_equal(got, want):
                bad += 1
                print(f"MISMATCH case {it} kernel {name}: B={B} Tx={Tx} Ty={Ty} kind={kind} neg={neg} tx={tx} ty={ty}", flush=True)
        if it % 20 == 19:
            print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    st = aligner_amd.read_status(dev)
    print(f"done: {n} cases, {bad} mismatches, status word {st}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
