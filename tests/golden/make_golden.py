#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (authoring container only).

Runs the reference's own code -- monotonic_align/core.pyx compiled into
oracle/_ref/core.so by `make -C oracle ref`, and monotonic_align/__init__.py
executed from /root/reference -- on seeded inputs and stores inputs + outputs as
small data fixtures.  Nothing of the reference's text is stored, only vectors.

    python tests/golden/make_golden.py        # rewrites the fixtures

The fixtures pin oracle/maxpath_oracle.{c,py} (tests/test_oracle.py) and are
the expected outputs of the HIP path in the `-m gpu` parity tests.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from aligner_amd import synth  # noqa: E402
from oracle import maxpath_oracle as O  # noqa: E402


def ref_core(value, tx, ty, ref, max_neg_val=None):
    v = np.ascontiguousarray(value, dtype=np.float32).copy()
    p = np.zeros(v.shape, np.int32)
    tx = np.ascontiguousarray(tx, dtype=np.int32)
    ty = np.ascontiguousarray(ty, dtype=np.int32)
    if max_neg_val is None:
        ref.maximum_path_c(p, v, tx, ty)
    else:
        ref.maximum_path_c(p, v, tx, ty, max_neg_val)
    return p, v


def small_kats(ref):
    """Known-answer cases: SURVEY 3.1 KATs + seeded random families."""
    rng = np.random.default_rng(20250117)
    cases = []

    def add(tag, value, tx, ty, neg=None):
        value = np.ascontiguousarray(value, dtype=np.float32)
        if value.ndim == 2:
            value = value[None]
        tx = np.atleast_1d(np.asarray(tx, np.int32))
        ty = np.atleast_1d(np.asarray(ty, np.int32))
        p, q = ref_core(value, tx, ty, ref, neg)
        # Q (the in-place mutated scores) is kept for the small cases only
        if q.size > 4096:
            q = np.zeros((0,), np.float32)
        cases.append(dict(tag=tag, value=value, tx=tx, ty=ty, path=p.astype(np.int8),
                          q=q, neg=np.float32(-1e9 if neg is None else neg)))

    add("arange_3x4", np.arange(1, 13).reshape(3, 4), 3, 4)
    add("zeros_3x6_ties", np.zeros((3, 6)), 3, 6)
    add("square_4x4", rng.standard_normal((4, 4)), 4, 4)
    add("tx1", rng.standard_normal((1, 9)), 1, 9)
    add("tx1_ty1", rng.standard_normal((1, 1)), 1, 1)
    add("padded_5x12_in_8x16", rng.standard_normal((8, 16)), 5, 12)
    add("neg_custom", rng.standard_normal((6, 20)), 6, 20, neg=-5.0)
    # denormals must survive (x86 SSE default keeps them; the GPU must too)
    den = (rng.integers(-40, 40, (7, 33)).astype(np.float32) * np.float32(1e-42))
    add("denormals_7x33", den, 7, 33)
    big = rng.standard_normal((5, 30)).astype(np.float32) * np.float32(1e37)
    add("overflow_to_inf_5x30", big, 5, 30)
    v = rng.standard_normal((6, 40)).astype(np.float32)
    v[rng.random(v.shape) < 0.10] = -np.inf
    add("neg_inf_6x40", v, 6, 40)
    v = rng.standard_normal((6, 40)).astype(np.float32)
    v[rng.random(v.shape) < 0.05] = np.nan
    v[rng.random(v.shape) < 0.05] = np.inf
    v[rng.random(v.shape) < 0.05] = -np.inf
    add("nan_inf_mix_6x40", v, 6, 40)
    # wave-boundary shapes for the kernel: rows around 63/64/65/126/127/128/129
    for tx in (63, 64, 65, 127, 128, 129):
        ty = tx + int(rng.integers(0, 40))
        add(f"gauss_rows{tx}", rng.standard_normal((tx, ty)), tx, ty)
        add(f"ties_rows{tx}", rng.integers(-1, 2, (tx, ty)), tx, ty)
    # tile-boundary frame counts: around multiples of 32
    for ty in (31, 32, 33, 63, 64, 65, 95, 96, 97):
        tx = int(rng.integers(1, min(ty, 40) + 1))
        add(f"gauss_cols{ty}", rng.standard_normal((tx, ty)), tx, ty)
    # random batches with ragged lengths inside a padded tensor
    for i in range(24):
        B = int(rng.integers(1, 5))
        Tx = int(rng.integers(1, 48))
        Ty = int(rng.integers(Tx, 120))
        kind = i % 3
        if kind == 0:
            val = rng.standard_normal((B, Tx, Ty))
        elif kind == 1:
            val = rng.integers(-3, 4, (B, Tx, Ty))
        else:
            val = -rng.random((B, Tx, Ty)) * 10
        ty = rng.integers(1, Ty + 1, B)
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty])
        add(f"ragged_{i}", val, tx, ty)
    # the forced diagonal move (core.pyx:34 `index == y`) must not depend on the scores:
    # NaN / -inf on and next to the diagonal, and running scores below max_neg_val
    for i, (tx, ty) in enumerate([(40, 60), (70, 90), (130, 150)]):
        v = rng.standard_normal((tx, ty)).astype(np.float32)
        v[rng.random(v.shape) < 0.08] = np.nan
        v[rng.random(v.shape) < 0.08] = -np.inf
        add(f"nan_on_diagonal_{i}", v, tx, ty)
        add(f"below_neg_{i}", np.full((tx, ty), -6e8, np.float32) + rng.integers(0, 3, (tx, ty)).astype(np.float32),
            tx, ty)
        add(f"all_nan_{i}", np.full((tx, ty), np.nan, np.float32), tx, ty)
    out = {"n": np.int32(len(cases)), "tags": np.array([c["tag"] for c in cases])}
    for i, c in enumerate(cases):
        for k in ("value", "tx", "ty", "path", "q", "neg"):
            out[f"c{i}_{k}"] = c[k]
    np.savez_compressed(os.path.join(HERE, "kat_small.npz"), **out)
    return len(cases)


def appendix_a(ref, refw):
    """hashes of path / durations for the BASELINE configs."""
    rec = {}
    keep = {}

    def run(tag, value, tx, ty, via_wrapper=False):
        B, Tx, Ty = value.shape
        if via_wrapper:
            mask = synth.prefix_mask(tx, ty, Tx, Ty)
            p = refw(torch.from_numpy(value), torch.from_numpy(mask)).numpy().astype(np.int32)
        else:
            p, _ = ref_core(value, tx, ty, ref)
        dur = p.sum(2).astype(np.int32)
        rec[tag] = dict(shape=[int(B), int(Tx), int(Ty)],
                        value_sha256=synth.sha256_of(value),
                        path_sha256=synth.sha256_of(p),
                        dur_sha256=synth.sha256_of(dur),
                        sum_tx=int(np.sum(tx)), sum_ty=int(np.sum(ty)),
                        dur0_16=[int(d) for d in dur[0][:16]])
        return p, dur

    full = lambda B, T: np.full(B, T, np.int32)  # noqa: E731
    v = synth.synth_value(*synth.CONFIGS["C1"])
    p, d = run("C1-fixed", v, full(4, 32), full(4, 128), via_wrapper=True)
    keep["c1_fixed_tok"] = p.argmax(1).astype(np.int16)
    p, d = run("C1-varlen", v, np.array([32, 20, 7, 1], np.int32),
               np.array([128, 100, 50, 9], np.int32), via_wrapper=True)
    keep["c1_varlen_dur"] = d.astype(np.int16)
    v = synth.synth_value(*synth.CONFIGS["C2"])
    p, d = run("C2-fixed", v, full(64, 200), full(64, 1000), via_wrapper=True)
    keep["c2_fixed_dur"] = d.astype(np.int16)
    tx, ty = synth.synth_lengths(64, 200, 500, 1000, 2)
    p, d = run("C2-varlen", v, tx, ty)
    keep["c2_varlen_dur"] = d.astype(np.int16)
    keep["c2_varlen_tx"], keep["c2_varlen_ty"] = tx, ty
    for s in range(8):
        v, tx, ty = synth.c4_shard(s)
        p, d = run(f"C4-shard{s}", v, tx, ty)
        if s == 0:
            keep["c4_s0_dur"] = d.astype(np.int16)
            keep["c4_s0_tx"], keep["c4_s0_ty"] = tx, ty
    v = synth.synth_value(*synth.CONFIGS["C5"])
    p, d = run("C5-longform", v, full(8, 500), full(8, 4000))
    keep["c5_dur"] = d.astype(np.int16)
    with open(os.path.join(HERE, "appendix_a.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "appendix_a_arrays.npz"), **keep)
    return rec


def wrapper_cases(refw):
    """Pins the L2 wrapper semantics (__init__.py:6-21): dtype/mask handling."""
    rng = np.random.default_rng(7)
    out = {}
    B, Tx, Ty = 3, 10, 24
    val = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([10, 6, 1], np.int32)
    ty = np.array([24, 17, 5], np.int32)
    mask = synth.prefix_mask(tx, ty, Tx, Ty)
    out["value"], out["mask"], out["tx"], out["ty"] = val, mask, tx, ty
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("f64", torch.float64)):
        r = refw(torch.from_numpy(val).to(dt), torch.from_numpy(mask).to(dt))
        assert r.dtype == dt
        out[f"path_{name}"] = r.to(torch.float32).numpy().astype(np.int8)
    r = refw(torch.from_numpy(val), torch.from_numpy(mask).bool())
    assert r.dtype == torch.float32
    out["path_boolmask"] = r.numpy().astype(np.int8)
    # a mask with interior zeros: lengths come from column 0 / row 0 only, the
    # zeros just zero the score (__init__.py:11,18-19)
    holes = mask.copy()
    holes[rng.random(holes.shape) < 0.15] = 0
    holes[:, :, 0] = mask[:, :, 0]
    holes[:, 0, :] = mask[:, 0, :]
    out["mask_holes"] = holes
    out["path_holes"] = refw(torch.from_numpy(val), torch.from_numpy(holes)).numpy().astype(np.int8)
    # garbage (non-zero, non-finite-free) scores in the padded region are ignored
    dirty = val.copy()
    dirty[mask == 0] = 1e6
    out["value_dirty"] = dirty
    out["path_dirty"] = refw(torch.from_numpy(dirty), torch.from_numpy(mask)).numpy().astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "wrapper_cases.npz"), **out)


def txgtty_cases(ref):
    """t_x > t_y (more text than frames): undefined in spirit, but the reference's OUTPUT is a function
    of in-bounds data only -- its forward band is empty (core.pyx:18), so the backtrack (core.pyx:32-35)
    compares RAW scores, and the one out-of-row read (y == 0) happens after the last path write.
    Utterance 0 of every batch is a valid one so that read stays inside the array."""
    rng = np.random.default_rng(4242)
    out = {}
    n = 0
    for (Tx, Ty) in [(4, 3), (6, 9), (12, 20), (40, 64), (70, 100), (9, 9)]:
        for kind in range(3):
            B = 4
            if kind == 0:
                v = rng.standard_normal((B, Tx, Ty))
            elif kind == 1:
                v = rng.integers(-1, 2, (B, Tx, Ty))           # ties
            else:
                v = np.zeros((B, Tx, Ty))
            v = v.astype(np.float32)
            ty = rng.integers(1, min(Tx - 1, Ty) + 1, B).astype(np.int32)
            tx = np.array([rng.integers(t + 1, Tx + 1) for t in ty], np.int32)
            tx[0], ty[0] = min(Tx, Ty), Ty                      # valid utterance in front
            if kind == 2:
                tx[1], ty[1] = Tx, 1                            # a single frame
            p, q = ref_core(v, tx, ty, ref)
            assert np.array_equal(q[1:], v[1:])                 # forward band empty: scores untouched
            out[f"c{n}_value"], out[f"c{n}_tx"], out[f"c{n}_ty"], out[f"c{n}_path"] = v, tx, ty, p.astype(np.int8)
            n += 1
    out["n"] = np.int32(n)
    np.savez_compressed(os.path.join(HERE, "kat_txgtty.npz"), **out)
    return n


def main():
    O.build(ref=True)
    ref = O.load_ref()
    refw = O.load_ref_wrapper()
    if ref is None or refw is None:
        raise SystemExit("reference not available: run in the authoring container")
    n = small_kats(ref)
    rec = appendix_a(ref, refw)
    wrapper_cases(refw)
    print(f"wrote {txgtty_cases(ref)} t_x > t_y cases")
    print(f"wrote {n} small KATs, {len(rec)} Appendix-A records")
    for k in ("C1-fixed", "C2-fixed", "C5-longform"):
        print(k, rec[k]["path_sha256"], rec[k]["dur0_16"])


if __name__ == "__main__":
    main()
