"""The oracle against the golden vectors made by the real reference
(tests/golden/make_golden.py) and against oracle/_ref when it is present."""
import numpy as np
import pytest

from aligner_amd import synth
from oracle import maxpath_oracle as O


def _run(value, tx, ty, neg=-1e9):
    v = value.copy()
    p = np.zeros(v.shape, np.int32)
    O.maximum_path_c(p, v, tx.copy(), ty.copy(), neg)
    return p, v


def test_kats_path_and_q_bit_exact(kats):
    assert len(kats) >= 50
    for c in kats:
        p, q = _run(c["value"], c["tx"], c["ty"], c["neg"])
        assert np.array_equal(p, c["path"].astype(np.int32)), c["tag"]
        if c["q"].size:
            # the in-place mutated scores, bit for bit (NaNs included)
            assert np.array_equal(q.view(np.uint32), c["q"].view(np.uint32)), c["tag"]


def test_tx_gt_ty_cases_match_reference(golden_dir):
    """More text than frames: the reference's forward band is empty, its backtrack runs on the raw scores
    (core.pyx:18,32-35); outputs of the compiled reference in tests/golden/kat_txgtty.npz."""
    import os
    z = np.load(os.path.join(golden_dir, "kat_txgtty.npz"))
    assert int(z["n"]) >= 12
    for i in range(int(z["n"])):
        p, q = _run(z[f"c{i}_value"], z[f"c{i}_tx"], z[f"c{i}_ty"])
        assert np.array_equal(p, z[f"c{i}_path"].astype(np.int32)), i


def test_restatement_under_asan_ubsan(kats):
    """the reference switches bounds checks off (core.pyx:7-8,38-39); the CPU build of
    its restatement runs the KATs under -fsanitize=address,undefined with every array its own exact-size
    heap block (oracle/sanitize_driver.c).  Any out-of-bounds access or UB aborts the driver."""
    import os
    import struct
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", odir, "sanitize"])
    blob = [struct.pack("<i", len(kats))]
    for c in kats:
        B, Tx, Ty = c["value"].shape
        blob += [struct.pack("<iiif", B, Tx, Ty, c["neg"]), c["tx"].astype("<i4").tobytes(),
                 c["ty"].astype("<i4").tobytes(), np.ascontiguousarray(c["value"], "<f4").tobytes()]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(odir, "sanitize_driver")], input=b"".join(blob), capture_output=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    off = 0
    for c in kats:
        n = c["value"].size
        p = np.frombuffer(r.stdout, "<i4", n, off).reshape(c["value"].shape); off += 4 * n
        q = np.frombuffer(r.stdout, "<u4", n, off).reshape(c["value"].shape); off += 4 * n
        assert np.array_equal(p, c["path"].astype(np.int32)), c["tag"]
        if c["q"].size:
            assert np.array_equal(q, c["q"].view(np.uint32)), c["tag"]
    assert off == len(r.stdout)


def test_inline_kats():
    # SURVEY 3.1
    v = np.arange(1, 13, dtype=np.float32).reshape(1, 3, 4)
    p, q = _run(v, np.array([3], np.int32), np.array([4], np.int32))
    assert p[0].tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]]
    assert q[0].tolist() == [[1, 3, 3, 4], [5, 7, 14, 8], [9, 10, 18, 30]]
    p, _ = _run(np.zeros((1, 3, 6), np.float32), np.array([3], np.int32), np.array([6], np.int32))
    assert p[0].tolist() == [[1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1]]


def test_column_sweep_formulation_matches(kats):
    """The kernel's formulation (frame-wise vector update + decision bits) == the loop nest."""
    for c in kats:
        for b in range(c["value"].shape[0]):
            tok, _ = O.column_sweep(c["value"][b], int(c["tx"][b]), int(c["ty"][b]), c["neg"])
            got = O.path_from_tok(tok, *c["value"].shape[1:])
            assert np.array_equal(got, c["path"][b].astype(np.int32)), c["tag"]


@pytest.mark.parametrize("tag", ["C1-fixed", "C1-varlen", "C2-fixed", "C2-varlen", "C4-shard0", "C4-shard3", "C4-shard7",
                                 "C5-longform"])
def test_appendix_a_hashes(appendix_a, tag):
    rec, _ = appendix_a
    r = rec[tag]
    B, Tx, Ty = r["shape"]
    if tag.startswith("C1"):
        v = synth.synth_value(*synth.CONFIGS["C1"])
        tx, ty = (np.full(B, Tx, np.int32), np.full(B, Ty, np.int32)) if tag == "C1-fixed" else \
            (np.array([32, 20, 7, 1], np.int32), np.array([128, 100, 50, 9], np.int32))
    elif tag.startswith("C2"):
        v = synth.synth_value(*synth.CONFIGS["C2"])
        tx, ty = (np.full(B, Tx, np.int32), np.full(B, Ty, np.int32)) if tag == "C2-fixed" else \
            synth.synth_lengths(64, 200, 500, 1000, 2)
    elif tag.startswith("C4"):
        v, tx, ty = synth.c4_shard(int(tag[-1]))
    else:
        v = synth.synth_value(*synth.CONFIGS["C5"])
        tx, ty = np.full(B, Tx, np.int32), np.full(B, Ty, np.int32)
    assert synth.sha256_of(v) == r["value_sha256"]
    assert int(tx.sum()) == r["sum_tx"] and int(ty.sum()) == r["sum_ty"]
    p, _ = _run(v, tx, ty)
    assert synth.sha256_of(p) == r["path_sha256"]
    dur = p.sum(2).astype(np.int32)
    assert synth.sha256_of(dur) == r["dur_sha256"]
    assert dur[0][:16].tolist() == r["dur0_16"]


def test_wrapper_restatement(golden_dir):
    """oracle.maximum_path (restating __init__.py:6-21) against the reference wrapper's outputs."""
    import os
    import torch
    z = np.load(os.path.join(golden_dir, "wrapper_cases.npz"))
    val, mask = torch.from_numpy(z["value"]), torch.from_numpy(z["mask"])
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("f64", torch.float64)):
        r = O.maximum_path(val.to(dt), mask.to(dt))
        assert r.dtype == dt and not r.requires_grad
        assert np.array_equal(r.float().numpy().astype(np.int8), z[f"path_{name}"])
    r = O.maximum_path(val, mask.bool())
    assert r.dtype == torch.float32
    assert np.array_equal(r.numpy().astype(np.int8), z["path_boolmask"])
    r = O.maximum_path(val, torch.from_numpy(z["mask_holes"]))
    assert np.array_equal(r.numpy().astype(np.int8), z["path_holes"])
    r = O.maximum_path(torch.from_numpy(z["value_dirty"]), mask)
    assert np.array_equal(r.numpy().astype(np.int8), z["path_dirty"])
    assert np.array_equal(z["path_dirty"], z["path_f32"])     # masked-out scores never matter


def test_oracle_rejects_what_the_reference_rejects():
    v = np.zeros((1, 2, 3), np.float32)
    p = np.zeros((1, 2, 3), np.int32)
    t = np.array([2], np.int32)
    u = np.array([3], np.int32)
    with pytest.raises(ValueError):
        O.maximum_path_c(p, v.astype(np.float64), t, u)
    with pytest.raises(ValueError):
        O.maximum_path_c(p, np.asfortranarray(np.zeros((2, 2, 3), np.float32))[:1], t, u)
    with pytest.raises(ValueError):
        O.maximum_path_c(p[0], v[0], t, u)


def test_against_compiled_reference_when_present():
    ref = O.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/core.so not built (reference tree absent)")
    rng = np.random.default_rng(5)
    for it in range(120):
        B, Tx = int(rng.integers(1, 4)), int(rng.integers(1, 40))
        Ty = int(rng.integers(Tx, 90))
        v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        if it % 3 == 1:
            v = np.round(v)                                        # heavy ties
        if it % 3 == 2:
            v[rng.random(v.shape) < 0.05] = -np.inf
            v[rng.random(v.shape) < 0.03] = np.nan
        ty = rng.integers(1, Ty + 1, B).astype(np.int32)
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
        p1, q1 = _run(v, tx, ty)
        q2 = v.copy()
        p2 = np.zeros(v.shape, np.int32)
        ref.maximum_path_c(p2, q2, tx, ty)
        assert np.array_equal(p1, p2)
        assert np.array_equal(q1.view(np.uint32), q2.view(np.uint32))
