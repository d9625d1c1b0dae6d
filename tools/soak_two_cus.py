#!/usr/bin/env python3
"""Soak of the two-workgroups-per-utterance form: random shapes with 253..504 text rows, ragged
lengths (some utterances without rows for the second workgroup, some with one), ties, long tokens, now and then a
non-finite score or 16-bit scores -- the path must equal the one-workgroup form's bit for bit, the status stay 0.

    python tools/soak_two_cus.py [seconds] [seed]
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
t0 = time.time(); it = 0; last = t0
while time.time() - t0 < secs:
    B = int(rng.integers(1, 13)); Tx = int(rng.integers(253, 505))
    Ty = int(rng.integers(Tx, Tx + int(rng.choice([40, 400, 1500, 3000]))))
    if rng.random() < 0.5: Ty = (Ty + 7) // 8 * 8                       # 16-byte rows: the wide kernels
    kind = it % 4
    v = (rng.standard_normal((B, Tx, Ty)) if kind < 2 else rng.integers(-2, 3, (B, Tx, Ty))).astype(np.float32)
    if kind == 1:
        for b in range(B): v[b, int(rng.integers(0, Tx)), :] += 3.0
    ty = rng.integers(max(Tx // 2, 1), Ty + 1, B).astype(np.int32); ty[0] = Ty
    tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32); tx[0] = Tx
    if B > 1 and rng.random() < 0.5: tx[1] = min(253, ty[1])
    if rng.random() < 0.15: v[int(rng.integers(0, B)), int(rng.integers(0, Tx)), int(rng.integers(0, Ty))] = rng.choice([np.inf, -np.inf, np.nan])
    dt = torch.float32 if rng.random() < 0.7 else (torch.bfloat16 if rng.random() < 0.5 else torch.float16)
    dv = torch.from_numpy(v).to(dt).to(dev); dtx = torch.from_numpy(tx).to(dev); dty = torch.from_numpy(ty).to(dev)
    r2 = aligner_amd.align(dv, dtx, dty, path_dtype=torch.int32, want_tok=True, cus_per_utterance=2)
    r1 = aligner_amd.align(dv, dtx, dty, path_dtype=torch.int32, want_tok=True, cus_per_utterance=1)
    torch.cuda.synchronize()
    ok = torch.equal(r1.path, r2.path) and torch.equal(r1.durations, r2.durations) and torch.equal(r1.tok, r2.tok)
    if not ok or aligner_amd.read_status(dev) != 0:
        print("MISMATCH", it, (B, Tx, Ty), dt, tx.tolist(), ty.tolist(), "status", aligner_amd.read_status(dev)); sys.exit(1)
    it += 1
    if time.time() - last > 20: print("...", it, "cases", flush=True); last = time.time()
print("ok:", it, "cases in", round(time.time() - t0, 1), "s")
