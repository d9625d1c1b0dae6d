#!/usr/bin/env python3
"""Randomized parity soak of the boundary search against the float64 oracle (development aid, not a test: minutes of
GPU + CPU time).  Shapes from tiny to long, windows from 1 to beyond the utterance, ragged batches (utterances with
fewer position segments than the launch), batches larger than the CU count, 16-bit energies, large energy scales
(windows far below the row's bulk: the exact sums), masked frames.

    python tools/soak_mobo.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd  # noqa: E402
from aligner_amd import mobo  # noqa: E402
from oracle import mobo_oracle as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0, bad, worst_la, worst_ga, worst_gr = time.time(), 0, 0.0, 0.0, 0.0
    for it in range(n):
        kind = int(rng.integers(0, 7))
        if kind == 0:                                           # tiny
            B, Tx, Ty, D = int(rng.integers(1, 6)), int(rng.integers(1, 9)), int(rng.integers(1, 40)), int(rng.integers(1, 12))
        elif kind == 1:                                         # many segments per utterance
            B, Tx, D = int(rng.integers(1, 4)), int(rng.integers(20, 200)), int(rng.choice([4, 7, 8, 16, 20, 32, 33, 64]))
            Ty = int(rng.integers(Tx, min(Tx * D, 2500) + 1))
        elif kind == 2:                                         # a window as long as (or longer than) the utterance
            B, Tx, Ty = int(rng.integers(1, 3)), int(rng.integers(1, 12)), int(rng.integers(30, 1800))
            D = int(rng.integers(max(1, Ty // 2), Ty + 50))
        elif kind == 3:                                         # more utterances than CUs: one segment each
            B, Tx, Ty, D = int(rng.integers(257, 400)), int(rng.integers(1, 6)), int(rng.integers(5, 60)), int(rng.integers(2, 20))
        elif kind == 4:                                         # several positions per thread
            B, Tx = 1, int(rng.integers(2, 8))
            Ty = int(rng.integers(1100, 3000)); D = int(rng.integers(Ty // 2, Ty))
        else:                                                   # mid-size
            B, Tx, D = int(rng.integers(1, 9)), int(rng.integers(5, 120)), int(rng.integers(2, 50))
            Ty = int(rng.integers(Tx, min(Tx * D, 1500) + 1))
        Ty = max(Ty, 1)
        if Ty > Tx * D:
            Ty = Tx * D
        if Ty < Tx:
            Tx = Ty
        scale = float(rng.choice([0.5, 2.0, 2.0, 8.0, 30.0]))
        e = (rng.standard_normal((B, Tx, Ty)) * scale).astype(np.float32)
        if rng.random() < 0.2:
            e[rng.random(e.shape) < 0.03] = -np.inf             # masked frames
        dt = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
        tx = np.array([Tx] + [int(rng.integers(max(1, -(-Ty // (2 * D))), Tx + 1)) for _ in range(B - 1)], np.int32)
        ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, B)], np.int32)
        ed = torch.from_numpy(e).to(dt)
        try:
            r = aligner_amd.boundary_search(ed.to(dev), torch.from_numpy(tx), torch.from_numpy(ty), D, want_gamma=True)
        except aligner_amd._lib.AlignerError as ex:              # the documented LDS limit (window x positions per segment)
            if ex.code == -33:
                continue
            raise
        # the gradient for random cotangents on log_alpha, on gamma or on both
        which = int(rng.integers(0, 3))
        g1 = rng.standard_normal((B, Tx, Ty)).astype(np.float32) if which != 1 else None
        g2 = rng.standard_normal((B, Tx, Ty)).astype(np.float32) if which != 0 else None
        gr = aligner_amd.boundary_search_backward(ed.to(dev), torch.from_numpy(tx), torch.from_numpy(ty), D, r.log_alpha,
                                                  None if g1 is None else torch.from_numpy(g1).to(dev),
                                                  None if g2 is None else torch.from_numpy(g2).to(dev))
        torch.cuda.synchronize()
        st = mobo.read_status(dev)
        gr = gr.cpu().numpy().astype(np.float64)
        la, ga = r.log_alpha.cpu().numpy().astype(np.float64), r.gamma.cpu().numpy().astype(np.float64)
        bnd, dur, sc = r.boundaries.cpu().numpy(), r.durations.cpu().numpy(), r.map_score.cpu().numpy()
        e64 = ed.float().numpy().astype(np.float64)
        ok = True
        msg = ""
        expect_st = 0
        for b in list(range(B)) if B <= 8 or it % 3 == 0 else list(range(6)):
            I, J = int(tx[b]), int(ty[b])
            want = M.boundary_search_fast(e64[b, :I, :J], D)
            if not np.isfinite(want["map_score"]):              # masked frames left no segmentation: zeros + ST_BAD_LENGTHS
                expect_st |= 1
                if dur[b].any() or bnd[b].any() or np.isfinite(sc[b]):
                    ok = False; msg = f"no-segmentation utterance b={b} not zeroed"; break
                if not np.all(np.isfinite(gr[b])):
                    ok = False; msg = f"no-segmentation utterance b={b}: gradient not finite"; break
                continue
            wg = M.boundary_search_backward(e64[b, :I, :J], D, None if g1 is None else g1[b, :I, :J].astype(np.float64),
                                            None if g2 is None else g2[b, :I, :J].astype(np.float64))
            gtol = (3e-3 * (1 + I / 100) * np.abs(wg).max() + 2e-5) * (4.0 if dt != torch.float32 or scale > 8 else 1.0) * max(1.0, scale / 2)
            dgr = np.abs(gr[b, :I, :J] - wg).max()
            worst_gr = max(worst_gr, dgr / gtol)
            if dgr > gtol or gr[b, I:].any() or gr[b, :, J:].any():
                ok = False; msg = f"gradient b={b} which={which} err={dgr:.2e} tol={gtol:.2e}"; break
            fin = np.isfinite(want["log_alpha"])
            if not np.array_equal(np.isfinite(la[b, :I, :J]), fin):
                ok = False; msg = f"finite pattern b={b}"; break
            dla = np.abs(la[b, :I, :J][fin] - want["log_alpha"][fin]).max() if fin.any() else 0.0
            dga = np.abs(ga[b, :I, :J] - want["gamma"]).max()
            worst_la, worst_ga = max(worst_la, dla / (2e-3 + 2e-5 * I)), max(worst_ga, dga / (2e-4 + 2e-6 * I))
            tol = 4.0 if dt != torch.float32 or scale > 8 else 1.0    # (the energies' own rounding is not the kernel's)
            if dla > tol * (2e-3 + 2e-5 * I) * max(1.0, scale / 2) or dga > tol * (2e-4 + 2e-6 * I) * max(1.0, scale / 2):
                ok = False; msg = f"values b={b} dla={dla:.2e} dga={dga:.2e}"; break
            bb = bnd[b, :I]
            if not (np.array_equal(np.diff(np.concatenate([[0], bb])), dur[b, :I]) and bb[-1] == J and dur[b, :I].min() >= 1
                    and dur[b, :I].max() <= D):
                ok = False; msg = f"segmentation b={b}"; break
            lp = M.sequence_log_prob(e64[b, :I, :J], D, bb)
            if np.isfinite(want["map_score"]) and not (lp >= want["map_score"] - (1e-3 + 1e-5 * I) * max(1.0, scale) and abs(sc[b] - lp) < (2e-3 + 2e-5 * I) * max(1.0, scale)):
                ok = False; msg = f"MAP b={b} lp={lp} want={want['map_score']} got={sc[b]}"; break
        if ok and (st & ~1) != 0:
            ok = False; msg = "status"
        if ok and B <= 8 and st != expect_st:
            ok = False; msg = f"status {st} expected {expect_st}"
        if not ok:
            bad += 1
            print(f"CASE {it} FAILED kind={kind} B={B} Tx={Tx} Ty={Ty} D={D} scale={scale} dt={dt} status={st}: {msg}", flush=True)
        if it % 10 == 9:
            print(f"{it + 1} cases, {bad} failures, worst la/ga/gradient error as a fraction of the test tolerance {worst_la:.2f} / {worst_ga:.2f} / {worst_gr:.2f}, "
                  f"{time.time() - t0:.0f}s", flush=True)
    print(f"done: {n} cases, {bad} failures", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
