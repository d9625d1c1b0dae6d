"""GPU parity tests: the HIP path (through the C ABI) against the oracle, the
golden vectors made by the real reference, and size-independent properties."""
import os

import numpy as np
import pytest
import torch

from aligner_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _oracle_path(value, tx, ty, neg=-1e9):
    from oracle import maxpath_oracle as O
    v = np.ascontiguousarray(value, dtype=np.float32).copy()
    p = np.zeros(v.shape, np.int32)
    O.maximum_path_c(p, v, np.ascontiguousarray(tx, np.int32), np.ascontiguousarray(ty, np.int32), neg)
    return p


def _hip(value, tx, ty, dev, **kw):
    import aligner_amd
    res = aligner_amd.align(torch.from_numpy(np.ascontiguousarray(value, np.float32)).to(dev),
                            torch.from_numpy(np.asarray(tx, np.int32)).to(dev),
                            torch.from_numpy(np.asarray(ty, np.int32)).to(dev),
                            path_dtype=torch.int32, want_tok=True, want_durations=True, **kw)
    torch.cuda.synchronize()
    return res.path.cpu().numpy(), res.tok.cpu().numpy(), res.durations.cpu().numpy()


def _check_consistency(path, tok, dur, tx, ty):
    B, Tx, Ty = path.shape
    assert np.array_equal(dur, path.sum(2))
    for b in range(B):
        assert np.all(tok[b, ty[b]:] == -1)
        if ty[b] > 0:
            assert np.array_equal(path[b].argmax(0)[:ty[b]], tok[b, :ty[b]])


KERNELS = {"wide": {}, "wide_retry_walk": {"no_prev_table": True}, "generic": {"force_generic": True},
           "wide_streamed_path": {"stream_path": True}, "wide_separate_expand": {"_test_flags": 4096}}


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_kats_bit_exact(kats, dev, kernel):
    """Both forward kernels (63-rows-per-wave pipelined -- with either form of the backtrack walk --, generic)."""
    import aligner_amd
    for c in kats:
        p, tok, dur = _hip(c["value"], c["tx"], c["ty"], dev, max_neg_val=c["neg"], **KERNELS[kernel])
        assert np.array_equal(p, c["path"].astype(np.int32)), c["tag"]
        _check_consistency(p, tok, dur, c["tx"], c["ty"])
    assert aligner_amd.read_status(dev) == 0


def _config(tag):
    if tag.startswith("C1"):
        v = synth.synth_value(*synth.CONFIGS["C1"])
        if tag == "C1-fixed":
            return v, np.full(4, 32, np.int32), np.full(4, 128, np.int32)
        return v, np.array([32, 20, 7, 1], np.int32), np.array([128, 100, 50, 9], np.int32)
    if tag.startswith("C2"):
        v = synth.synth_value(*synth.CONFIGS["C2"])
        if tag == "C2-fixed":
            return v, np.full(64, 200, np.int32), np.full(64, 1000, np.int32)
        return (v,) + synth.synth_lengths(64, 200, 500, 1000, 2)
    if tag.startswith("C4"):
        return synth.c4_shard(int(tag[-1]))
    v = synth.synth_value(*synth.CONFIGS["C5"])
    return v, np.full(8, 500, np.int32), np.full(8, 4000, np.int32)


@pytest.mark.parametrize("tag", ["C1-fixed", "C1-varlen", "C2-fixed", "C2-varlen"] +
                         [f"C4-shard{s}" for s in range(8)] + ["C5-longform"])
def test_baseline_configs_match_reference_hashes(appendix_a, dev, tag):
    """Full-size BASELINE.json configs: sha256 of the int32 path and of the durations
    equal the values captured from the real reference (tests/golden/appendix_a.json)."""
    rec, _ = appendix_a
    v, tx, ty = _config(tag)
    p, tok, dur = _hip(v, tx, ty, dev)
    assert synth.sha256_of(p) == rec[tag]["path_sha256"]
    assert synth.sha256_of(dur.astype(np.int32)) == rec[tag]["dur_sha256"]
    _check_consistency(p, tok, dur, tx, ty)


@pytest.mark.parametrize("kernel", ["wide_retry_walk", "generic"])
def test_other_kernels_agree_on_c2(appendix_a, dev, kernel):
    rec, _ = appendix_a
    for tag in ("C2-fixed", "C2-varlen"):
        v, tx, ty = _config(tag)
        p, _, dur = _hip(v, tx, ty, dev, **KERNELS[kernel])
        assert synth.sha256_of(p) == rec[tag]["path_sha256"]


def test_random_shapes_vs_oracle(dev):
    rng = np.random.default_rng(1234)
    for it in range(40):
        B = int(rng.integers(1, 6))
        Tx = int(rng.integers(1, 300))
        Ty = int(rng.integers(Tx, Tx + 400))
        kind = it % 4
        if kind == 0:
            v = rng.standard_normal((B, Tx, Ty))
        elif kind == 1:
            v = rng.integers(-2, 3, (B, Tx, Ty))              # heavy ties
        elif kind == 2:
            v = -rng.random((B, Tx, Ty)) * 20
        else:
            v = rng.standard_normal((B, Tx, Ty))
            v[rng.random(v.shape) < 0.03] = -np.inf
            v[rng.random(v.shape) < 0.01] = np.nan
            v[rng.random(v.shape) < 0.01] = np.inf
        v = v.astype(np.float32)
        ty = rng.integers(1, Ty + 1, B).astype(np.int32)
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
        want = _oracle_path(v, tx, ty)
        for kernel, kw in KERNELS.items():
            p, tok, dur = _hip(v, tx, ty, dev, **kw)
            assert np.array_equal(p, want), (it, kernel, B, Tx, Ty)
            _check_consistency(p, tok, dur, tx, ty)


@pytest.mark.parametrize("shape", [(3, 200, 1004), (2, 64, 100), (4, 127, 420), (1, 253, 996), (5, 300, 332)])
def test_vector_loader_tail_tiles(dev, shape):
    """16-byte loader path (Ty % 4 == 0) with a last tile that runs past the row end and, for the last
    rows of an utterance, past its [Tx,Ty] block: the loaders' buffer loads must neither fault nor let
    what they read there (the next row, the next utterance, zeros) reach the result."""
    B, Tx, Ty = shape
    rng = np.random.default_rng(Tx * 1000 + Ty)
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    v[-1] = np.inf                                   # poison everything the last utterance must not read ...
    tx = np.full(B, Tx, np.int32)
    ty = np.full(B, Ty, np.int32)
    tx[-1], ty[-1] = max(1, Tx // 3), max(Tx // 3, Ty // 2)
    v[-1, :tx[-1], :ty[-1]] = rng.standard_normal((tx[-1], ty[-1])).astype(np.float32)   # ... except its own cells
    want = _oracle_path(v, tx, ty)
    for kw in ({}, {"force_generic": True}):
        p, tok, dur = _hip(v, tx, ty, dev, **kw)
        assert np.array_equal(p, want), kw
        _check_consistency(p, tok, dur, tx, ty)


def test_long_utterances_windowed_backtrack(dev):
    """T_mel > 2048: more than 64 tiles of decision words, so the backtrack runs over several windows read
    back from the workspace.  Ragged lengths, ties and long tokens (rows that span a window boundary)."""
    rng = np.random.default_rng(77)
    for it, (B, Tx, Ty) in enumerate([(3, 120, 2500), (2, 300, 4100), (4, 64, 3000), (2, 500, 2080), (3, 30, 6200)]):
        if it % 2 == 0:
            v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        else:
            v = rng.integers(-2, 3, (B, Tx, Ty)).astype(np.float32)       # heavy ties
        # a few rows that are strongly preferred for thousands of frames: tokens longer than a window
        for b in range(B):
            r = int(rng.integers(0, Tx))
            v[b, r, :] += 3.0
        ty = rng.integers(Ty // 2, Ty + 1, B).astype(np.int32)
        ty[0] = Ty
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
        tx[0] = Tx
        want = _oracle_path(v, tx, ty)
        for kw in ({}, {"force_generic": True}):
            p, tok, dur = _hip(v, tx, ty, dev, **kw)
            assert np.array_equal(p, want), (it, kw)
            _check_consistency(p, tok, dur, tx, ty)


def test_two_streams_do_not_share_workspaces(dev):
    """align() / soft_attention() are asynchronous on the current stream and keep live state (token starts,
    decision words) in a workspace between their launches: calls in flight on two streams of one device must
    not share one.  Two different batches, interleaved on two streams, each checked against the oracle."""
    import aligner_amd
    rng = np.random.default_rng(31)
    shapes = [(48, 200, 1000), (64, 150, 800)]
    vals = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
    lens = [synth.synth_lengths(sh[0], sh[1], sh[2] // 2, sh[2], 5 + i) for i, sh in enumerate(shapes)]
    want = [_oracle_path(v, tx, ty) for v, (tx, ty) in zip(vals, lens)]
    dv = [torch.from_numpy(v).to(dev) for v in vals]
    dl = [(torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)) for tx, ty in lens]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    for rep in range(6):
        res = [None, None]
        for i in (0, 1) if rep % 2 == 0 else (1, 0):
            with torch.cuda.stream(streams[i]):
                res[i] = aligner_amd.align(dv[i], dl[i][0], dl[i][1], path_dtype=torch.int32, want_tok=True)
        torch.cuda.synchronize()
        for i in (0, 1):
            assert np.array_equal(res[i].path.cpu().numpy(), want[i]), (rep, i)
            assert np.array_equal(res[i].durations.cpu().numpy(), want[i].sum(2)), (rep, i)
    # same for the similarity front end (prepared text operand in its workspace)
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(8)
    ks = [torch.randn(16, 80, 300, generator=g), torch.randn(24, 80, 260, generator=g)]
    qs = [torch.randn(16, 80, 900, generator=g), torch.randn(24, 80, 700, generator=g)]
    wl = [torch.cat([S.soft_attention(k[b:b + 1], q[b:b + 1])[0] for b in range(k.shape[0])]) for k, q in zip(ks, qs)]
    dk, dq = [k.to(dev) for k in ks], [q.to(dev) for q in qs]
    torch.cuda.synchronize()
    for rep in range(4):
        out = [None, None]
        for i in (0, 1) if rep % 2 == 0 else (1, 0):
            with torch.cuda.stream(streams[i]):
                out[i] = aligner_amd.soft_attention(dk[i], dq[i])[0]
        torch.cuda.synchronize()
        for i in (0, 1):
            assert (out[i].cpu() - wl[i]).abs().max().item() < 1e-4, (rep, i)


def test_wide_text_uses_generic_path(dev):
    """Tx > 512 is outside the pipelined kernel; the generic kernel must take over."""
    rng = np.random.default_rng(9)
    v = rng.standard_normal((2, 700, 900)).astype(np.float32)
    tx = np.array([700, 513], np.int32)
    ty = np.array([900, 640], np.int32)
    p, tok, dur = _hip(v, tx, ty, dev)
    assert np.array_equal(p, _oracle_path(v, tx, ty))


def test_path_invariants_full_size(dev):
    """Size-independent properties at the headline size on gaussian scores (no fixture)."""
    g = torch.Generator(device="cpu").manual_seed(3)
    v = torch.randn(64, 200, 1000, generator=g).numpy()
    tx, ty = synth.synth_lengths(64, 200, 500, 1000, 7)
    p, tok, dur = _hip(v, tx, ty, dev)
    for b in range(64):
        t = tok[b, :ty[b]]
        assert t[0] == 0 and t[-1] == tx[b] - 1
        d = np.diff(t)
        assert np.all((d == 0) | (d == 1))                     # monotone, one row at a time
        assert np.all(dur[b, :tx[b]] >= 1) and dur[b].sum() == ty[b]
        assert p[b, tx[b]:, :].sum() == 0 and p[b, :, ty[b]:].sum() == 0
    assert np.all(p.sum(1)[np.arange(1000)[None, :] < ty[:, None]] == 1)
    # the chosen path's score is the DP optimum: compare with the oracle's path score
    want = _oracle_path(v, tx, ty)
    assert np.array_equal(p, want)


def test_strict_mask_and_wrapper_semantics(golden_dir, dev):
    """maximum_path(value, mask): dtype/device contract of __init__.py:6-21 on GPU tensors,
    against outputs of the reference wrapper (tests/golden/wrapper_cases.npz)."""
    import aligner_amd
    z = np.load(os.path.join(golden_dir, "wrapper_cases.npz"))
    val, mask = torch.from_numpy(z["value"]).to(dev), torch.from_numpy(z["mask"]).to(dev)
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("f64", torch.float64)):
        vin = val.to(dt).requires_grad_(True)
        r = aligner_amd.maximum_path(vin, mask.to(dt))
        assert r.dtype == dt and r.device == vin.device and not r.requires_grad
        assert np.array_equal(r.float().cpu().numpy().astype(np.int8), z[f"path_{name}"])
    r = aligner_amd.maximum_path(val, mask.bool())
    assert r.dtype == torch.float32
    assert np.array_equal(r.cpu().numpy().astype(np.int8), z["path_boolmask"])
    # interior zeros in the mask change the scores (strict multiply, __init__.py:11)
    r = aligner_amd.maximum_path(val, torch.from_numpy(z["mask_holes"]).to(dev))
    assert np.array_equal(r.cpu().numpy().astype(np.int8), z["path_holes"])
    # garbage in the masked-out region is ignored, with and without the multiply
    dirty = torch.from_numpy(z["value_dirty"]).to(dev)
    for prefix in (False, True):
        r = aligner_amd.maximum_path(dirty, mask, mask_is_prefix=prefix)
        assert np.array_equal(r.cpu().numpy().astype(np.int8), z["path_dirty"])
    # bf16 (extension: the reference raises TypeError): same path as the fp32 upcast
    r = aligner_amd.maximum_path(val.bfloat16(), mask.bfloat16())
    want = aligner_amd.maximum_path(val.bfloat16().float(), mask)
    assert r.dtype == torch.bfloat16 and torch.equal(r.float(), want)
    # caller's tensors untouched
    assert torch.equal(val.cpu(), torch.from_numpy(z["value"]))
    # CPU tensors are staged through the GPU and come back on the CPU
    r = aligner_amd.maximum_path(torch.from_numpy(z["value"]), torch.from_numpy(z["mask"]))
    assert r.device.type == "cpu" and np.array_equal(r.numpy().astype(np.int8), z["path_f32"])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Tx,Ty", [(5, 200, 1000), (3, 70, 263), (2, 300, 808), (9, 33, 132), (2, 500, 4000)])
def test_strict_mask_verified_on_the_device(dev, request, dt, B, Tx, Ty):
    """The drop-in's strict mask (value * mask, __init__.py:11) without the second stream where it is a no-op:
    mask_verify_kernel checks, per utterance, that the mask is ONE on every score the search reads and the search then
    skips it.  Prefix rectangles (ragged, garbage scores outside them), one utterance whose mask has a hole, one whose
    rectangle holds a value that is not one, one in the reference's t_x > t_y mode: each against the multiply done in
    torch + the oracle, and bit for bit against the kernels with the verification switched off."""
    import aligner_amd
    from aligner_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B * 7 + Tx + Ty)
    v = torch.from_numpy(rng.standard_normal((B, Tx, Ty)).astype(np.float32) * 3).to(dt)
    ty = rng.integers(Tx, Ty + 1, B).astype(np.int32)
    tx = np.array([rng.integers(1, Tx + 1) for _ in ty], np.int32)
    tx[0], ty[0] = Tx, Ty
    m = torch.from_numpy(synth.prefix_mask(tx, ty, Tx, Ty)).to(dt)
    v = v + (1 - m) * 1e4                                        # garbage where the mask is zero: must not matter
    cases = {"prefix": m.clone()}
    holes = m.clone()
    holes[B - 1, tx[B - 1] // 2, ty[B - 1] // 2] = 0             # a hole inside the last utterance's rectangle
    cases["hole"] = holes
    scaled = m.clone()
    scaled[0, Tx - 1, Ty - 1] = 0.5                              # the first utterance's last cell is not one
    cases["not one"] = scaled
    if B > 1 and Tx > 4:
        degen = m.clone()
        degen[1] = 0
        degen[1, :4, :2] = 1                                     # t_x = 4 > t_y = 2: the reference's raw-score walk
        cases["t_x > t_y"] = degen
    request.addfinalizer(lambda: (lib.aligner_debug_set_option(b"maxpath_no_mask_verify", 0),
                                  lib.aligner_debug_set_option(b"maxpath_no_optimistic_mask", 0)))
    for name, mask in cases.items():
        # the library's choice (an optimistic search beside the verification on the zero workgroups + a redo of the flagged
        # utterances, where the launch has zero workgroups), the verification as a pass in front of the search, and the
        # multiply always: one answer.  Twice in a row: the redo erases the optimistic search's ones from the dense path
        got = aligner_amd.maximum_path(v.to(dev), mask.to(dev))
        again = aligner_amd.maximum_path(v.to(dev), mask.to(dev))
        assert lib.aligner_debug_set_option(b"maxpath_no_optimistic_mask", 1) == 0
        front = aligner_amd.maximum_path(v.to(dev), mask.to(dev))
        lib.aligner_debug_set_option(b"maxpath_no_optimistic_mask", 0)
        assert lib.aligner_debug_set_option(b"maxpath_no_mask_verify", 1) == 0
        mul = aligner_amd.maximum_path(v.to(dev), mask.to(dev))
        lib.aligner_debug_set_option(b"maxpath_no_mask_verify", 0)
        torch.cuda.synchronize()
        assert got.dtype == dt and torch.equal(got, mul) and torch.equal(again, mul) and torch.equal(front, mul), name
        # durations and token indices of the redone utterances too (align() with the mask multiplied in)
        ra = aligner_amd.align(v.to(dev), mask=mask.to(dev), strict_mask=True, want_tok=True, compat_tx_gt_ty=True)
        torch.cuda.synchronize()
        assert torch.equal(ra.path, mul), name
        assert torch.equal(ra.durations, mul.float().sum(2).to(torch.int32)), name
        if name != "t_x > t_y":
            lyv = mask[:, 0, :].float().sum(1).int()
            tokw = torch.where(torch.arange(Ty)[None, :] < lyv[:, None], mul.float().argmax(1).cpu().to(torch.int32), -1)
            assert torch.equal(ra.tok.cpu(), tokw), name
        if name != "t_x > t_y":
            lx = mask[:, :, 0].float().sum(1).int().numpy()
            ly = mask[:, 0, :].float().sum(1).int().numpy()
            want = _oracle_path((v * mask).float().numpy(), lx, ly)
            assert np.array_equal(got.float().cpu().numpy().astype(np.int32), want), name
    # lengths given by the caller, the mask only multiplied in
    r = aligner_amd.align(v.to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev), mask=cases["hole"].to(dev),
                          strict_mask=True, path_dtype=torch.int32)
    want = _oracle_path((v * cases["hole"]).float().numpy(), tx, ty)
    assert np.array_equal(r.path.cpu().numpy(), want)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Tx,Ty", [(5, 200, 1000), (3, 70, 264), (2, 300, 808), (9, 33, 136), (2, 130, 999), (1, 1, 40)])
def test_scores_with_a_row_pitch(dev, dt, B, Tx, Ty):
    """aligner_maxpath_ld: scores whose rows start on whole 128-byte lines (softattn.pitched_logp -- the layout the pipeline
    hands from the similarity kernel to the search) give the path, durations and token indices of the contiguous layout,
    whatever the padding columns hold (here: NaNs).  Ragged lengths, the eight-wave form (300 rows), an odd T_mel (the
    generic loaders), a NaN among the scores (the exact sweep), the generic kernel, the reference's t_x > t_y walk."""
    import aligner_amd
    from aligner_amd.softattn import pitched_logp
    rng = np.random.default_rng(B * 11 + Tx + Ty)
    v = torch.from_numpy(rng.standard_normal((B, Tx, Ty)).astype(np.float32) * 3).to(dt)
    ty = rng.integers(max(Tx, 1), Ty + 1, B).astype(np.int32)
    tx = np.array([rng.integers(1, Tx + 1) for _ in ty], np.int32)
    tx[0], ty[0] = Tx, Ty
    if B > 2:
        v[2, min(3, Tx - 1), 5] = float("nan")
    pv = pitched_logp(B, Tx, Ty, dev, dt)
    ld = pv.stride(1) if Tx > 1 else pv.stride(0)
    torch.as_strided(pv, (B, Tx, ld), (Tx * ld, ld, 1)).fill_(float("nan"))
    pv.copy_(v.to(dev))
    txd, tyd = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    for kw in ({}, {"force_generic": True}, {"want_tok": True, "path_dtype": torch.int32}):
        a = aligner_amd.align(v.to(dev), txd, tyd, **kw)
        b = aligner_amd.align(pv, txd, tyd, **kw)
        torch.cuda.synchronize()
        assert torch.equal(a.path, b.path) and torch.equal(a.durations, b.durations), kw
        if kw.get("want_tok"):
            assert torch.equal(a.tok, b.tok)
    want = _oracle_path(v.float().numpy(), tx, ty)
    assert np.array_equal(b.path.cpu().numpy(), want)
    if Tx > 4 and B > 1:
        tx2, ty2 = tx.copy(), ty.copy()
        tx2[1], ty2[1] = 4, 2                                    # the reference's raw-score walk from row t_x - 1
        a = aligner_amd.align(v.to(dev), torch.from_numpy(tx2).to(dev), torch.from_numpy(ty2).to(dev), compat_tx_gt_ty=True)
        b = aligner_amd.align(pv, torch.from_numpy(tx2).to(dev), torch.from_numpy(ty2).to(dev), compat_tx_gt_ty=True)
        assert torch.equal(a.path, b.path) and torch.equal(a.durations, b.durations)
    # a mask comes with the reference's contiguous tensors: a pitched value is copied then, never misread -- and the drop-in
    # call refuses it as the reference's memoryview does (core.c:27843)
    m = torch.from_numpy(synth.prefix_mask(tx, ty, Tx, Ty)).to(dt).to(dev)
    a = aligner_amd.align(v.to(dev), mask=m, strict_mask=True)
    b = aligner_amd.align(pv, mask=m, strict_mask=True)
    assert torch.equal(a.path, b.path)
    if not pv.is_contiguous():
        with pytest.raises(ValueError, match="C-contiguous"):
            aligner_amd.maximum_path(pv, m)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_sixteen_bit_scores_read_directly(appendix_a, dev, dt):
    """bf16 / fp16 scores go straight into the DP's loaders (no up-cast pass): same path as the fp32 up-cast.
    BASELINE config C5 ("bf16 similarity + int32 path", [8,500,4000], scores exactly representable in bf16 and
    fp16) against the reference's hash; ragged random batches, an odd T_mel (generic kernel) and the strict
    mask product -- rounded in the tensors' dtype, as torch's value * mask is -- against the oracle."""
    import aligner_amd
    rec, _ = appendix_a
    v = torch.from_numpy(synth.synth_value(*synth.CONFIGS["C5"]))
    assert torch.equal(v.to(dt).float(), v)                      # exactly representable: the golden hash applies
    r = aligner_amd.align(v.to(dt).to(dev), torch.full((8,), 500, dtype=torch.int32, device=dev),
                          torch.full((8,), 4000, dtype=torch.int32, device=dev), path_dtype=torch.int32)
    torch.cuda.synchronize()
    assert synth.sha256_of(r.path.cpu().numpy()) == rec["C5-longform"]["path_sha256"]
    assert synth.sha256_of(r.durations.cpu().numpy().astype(np.int32)) == rec["C5-longform"]["dur_sha256"]
    rng = np.random.default_rng(5)
    for (B, Tx, Ty) in [(5, 200, 1000), (3, 70, 264), (2, 300, 808), (3, 33, 131), (2, 130, 999)]:
        v = torch.from_numpy(rng.standard_normal((B, Tx, Ty)).astype(np.float32) * 3).to(dt)
        if Tx == 70:
            v[0, 3, 5] = float("nan")                            # the exact-sweep fallback with 16-bit scores
            v[1, 10, 40] = float("inf")
        ty = rng.integers(Tx, Ty + 1, B).astype(np.int32)
        tx = np.array([rng.integers(1, Tx + 1) for _ in ty], np.int32)
        tx[0], ty[0] = Tx, Ty
        want = _oracle_path(v.float().numpy(), tx, ty)
        for kw in ({}, {"no_prev_table": True}, {"force_generic": True}):
            r = aligner_amd.align(v.to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                                  path_dtype=torch.int32, **kw)
            torch.cuda.synchronize()
            assert np.array_equal(r.path.cpu().numpy(), want), (B, Tx, Ty, kw)
        # strict mask in the scores' dtype, with values that are not 0/1: the product is rounded in that dtype
        mask = torch.from_numpy(synth.prefix_mask(tx, ty, Tx, Ty)).to(dt) * torch.from_numpy(
            rng.uniform(0.5, 1.5, (B, Tx, Ty)).astype(np.float32)).to(dt)
        mask[:, :, 0] = torch.from_numpy(synth.prefix_mask(tx, ty, Tx, Ty))[:, :, 0].to(dt)   # lengths are read from
        mask[:, 0, :] = torch.from_numpy(synth.prefix_mask(tx, ty, Tx, Ty))[:, 0, :].to(dt)   # column 0 / row 0
        want = _oracle_path((v * mask).float().numpy(), tx, ty)
        got = aligner_amd.maximum_path(v.to(dev), mask.to(dev))
        assert got.dtype == dt
        assert np.array_equal(got.float().cpu().numpy().astype(np.int32), want), (B, Tx, Ty, "strict")


def test_maximum_path_c_numpy_boundary(kats, dev):
    """core.pyx:40 signature on host buffers, through aligner_maxpath_host_f32: the path AND the scores the
    reference leaves behind -- `values` comes back as the running score Q inside the band (core.pyx:30), bit for bit
    against the real reference's own mutated buffers; write_q=False leaves it untouched."""
    import aligner_amd
    checked_q = 0
    for c in kats:
        ok = bool(np.all((c["tx"] >= 1) & (c["tx"] <= c["ty"])))          # (the rest: test_degenerate_lengths)
        if not ok:
            continue
        p = np.zeros(c["value"].shape, np.int32)
        v = c["value"].copy()
        aligner_amd.maximum_path_c(p, v, c["tx"].copy(), c["ty"].copy(), c["neg"])
        assert np.array_equal(p, c["path"].astype(np.int32)), c["tag"]
        if c["q"].size:
            # every finite or infinite score bit for bit; NaNs in the same cells (their sign and payload are the
            # adding hardware's: x86 makes 0xFFC00000 of inf - inf, gfx950 0x7FC00000 -- not the algorithm's)
            nan = np.isnan(c["q"])
            assert np.array_equal(np.isnan(v), nan), c["tag"]
            assert np.array_equal(v.view(np.uint32)[~nan], c["q"].view(np.uint32)[~nan]), c["tag"]
            checked_q += 1
        p2 = np.zeros(c["value"].shape, np.int32)
        v2 = c["value"].copy()
        aligner_amd.maximum_path_c(p2, v2, c["tx"].copy(), c["ty"].copy(), c["neg"], write_q=False)
        assert np.array_equal(p2, p) and np.array_equal(v2.view(np.uint32), c["value"].view(np.uint32)), c["tag"]
    assert checked_q >= 30
    # a full-size batch: Q against the C restatement (itself pinned to those fixtures)
    from oracle import maxpath_oracle as O
    rng = np.random.default_rng(3)
    B, Tx, Ty = 3, 300, 900
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    ty = np.array([Ty, 700, 300], np.int32)
    tx = np.array([Tx, 120, 300], np.int32)
    want_p, want_q = np.zeros(v.shape, np.int32), v.copy()
    O.maximum_path_c(want_p, want_q, tx, ty, -1e9)
    p = np.zeros(v.shape, np.int32)
    aligner_amd.maximum_path_c(p, v, tx, ty, -1e9)
    assert np.array_equal(p, want_p) and np.array_equal(v.view(np.uint32), want_q.view(np.uint32))
    with pytest.raises(ValueError):
        aligner_amd.maximum_path_c(np.zeros((1, 4, 3), np.int32), np.zeros((1, 4, 3), np.float32),
                                   np.array([4], np.int32), np.array([3], np.int32))


def test_degenerate_lengths(dev):
    """t_y == 0 -> empty path; t_x > t_y -> the reference's all-last-row result (compat) or a
    status bit; t_x == 0 -> status bit, zero path (the reference writes out of bounds)."""
    import aligner_amd
    rng = np.random.default_rng(2)
    v = torch.from_numpy(rng.standard_normal((4, 6, 9)).astype(np.float32)).to(dev)
    tx = torch.tensor([3, 4, 0, 2], dtype=torch.int32, device=dev)
    ty = torch.tensor([0, 3, 5, 9], dtype=torch.int32, device=dev)
    r = aligner_amd.align(v, tx, ty, path_dtype=torch.int32, want_tok=True, compat_tx_gt_ty=True)
    torch.cuda.synchronize()
    p = r.path.cpu().numpy()
    assert p[0].sum() == 0
    # t_x > t_y: the reference backtracks on the raw scores (forward band empty), see test_tx_gt_ty_*
    assert np.array_equal(p[1], _oracle_path(v[0:2].cpu().numpy(), np.array([3, 4]), np.array([0, 3]))[1])
    assert p[2].sum() == 0
    assert aligner_amd.read_status(dev) & 1
    want = _oracle_path(v[3:].cpu().numpy(), np.array([2]), np.array([9]))
    assert np.array_equal(p[3], want[0])
    r = aligner_amd.align(v, tx, ty, path_dtype=torch.int32)
    assert r.path[1].sum().item() == 0 and (aligner_amd.read_status(dev) & 1)


def test_tx_gt_ty_matches_reference_outputs(golden_dir, dev):
    """More text than frames: outputs of the compiled reference (tests/golden/kat_txgtty.npz, made by
    make_golden.py) for the compat flag the drop-in wrapper sets; with and without the strict mask."""
    import aligner_amd
    z = np.load(os.path.join(golden_dir, "kat_txgtty.npz"))
    for i in range(int(z["n"])):
        v, tx, ty, want = z[f"c{i}_value"], z[f"c{i}_tx"], z[f"c{i}_ty"], z[f"c{i}_path"].astype(np.int32)
        for kw in ({}, {"force_generic": True}):
            p, tok, dur = _hip(v, tx, ty, dev, compat_tx_gt_ty=True, **kw)
            assert np.array_equal(p, want), (i, kw)
            assert np.array_equal(dur, want.sum(2))
            for b in range(len(tx)):
                assert np.array_equal(tok[b, :ty[b]], want[b].argmax(0)[:ty[b]]) and np.all(tok[b, ty[b]:] == -1)
        B, Tx, Ty = v.shape
        mask = synth.prefix_mask(tx, ty, Tx, Ty)
        r = aligner_amd.maximum_path(torch.from_numpy(v).to(dev), torch.from_numpy(mask).to(dev))
        assert np.array_equal(r.cpu().numpy().astype(np.int32), want), i
    assert aligner_amd.read_status(dev) == 0


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.int32,
                                torch.uint8, torch.int64])
def test_expand_dtypes(dev, dt):
    import aligner_amd
    v = synth.synth_value(3, 17, 50, 99)
    tx = np.array([17, 5, 9], np.int32)
    ty = np.array([50, 33, 9], np.int32)
    want = _oracle_path(v, tx, ty)
    r = aligner_amd.align(torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                          path_dtype=dt)
    assert r.path.dtype == dt
    assert np.array_equal(r.path.to(torch.int32).cpu().numpy(), want)
    # the same through the non-temporal stores of ALIGNER_F_STREAM_PATH (16-byte-aligned rows: Ty = 52)
    v4 = synth.synth_value(3, 17, 52, 98)
    ty4 = np.array([52, 33, 9], np.int32)
    r = aligner_amd.align(torch.from_numpy(v4).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty4).to(dev),
                          path_dtype=dt, stream_path=True)
    assert r.path.dtype == dt and np.array_equal(r.path.to(torch.int32).cpu().numpy(), _oracle_path(v4, tx, ty4))
    # the caller's zeros + the search kernel's own ones (ALIGNER_F_PATH_PREZEROED), every kernel form
    for kw in ({}, {"force_generic": True}, {"no_prev_table": True}):
        out = torch.zeros((3, 17, 52), dtype=dt, device=dev)
        r = aligner_amd.align(torch.from_numpy(v4).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty4).to(dev),
                              out_path=out, out_path_is_zero=True, **kw)
        assert r.path is out and np.array_equal(out.to(torch.int32).cpu().numpy(), _oracle_path(v4, tx, ty4)), kw
    # odd Ty exercises the unaligned (scalar) load/store paths
    v = synth.synth_value(2, 9, 37, 5)
    r = aligner_amd.align(torch.from_numpy(v).to(dev), torch.tensor([9, 3], dtype=torch.int32, device=dev),
                          torch.tensor([37, 20], dtype=torch.int32, device=dev), path_dtype=dt)
    assert np.array_equal(r.path.to(torch.int32).cpu().numpy(), _oracle_path(v, [9, 3], [37, 20]))


def test_sharded_product_path_single_rank(appendix_a, dev):
    """The multi-GPU entry point with the real HIP align_fn on a 1-rank RCCL group:
    durations of C4 shard 0 equal the reference's (tests/golden/appendix_a_arrays.npz)."""
    import torch.distributed as dist
    from aligner_amd import sharded
    _, arrays = appendix_a
    v, tx, ty = synth.c4_shard(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        vd = torch.from_numpy(v).to(dev)
        got = sharded.sharded_align(lambda idx: vd[torch.as_tensor(idx, device=dev)], tx, ty, v.shape[1], device=dev)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert np.array_equal(got.cpu().numpy(), arrays["c4_s0_dur"].astype(np.int32))


def test_token_index_on_a_very_long_mel_axis(dev):
    """tok[b, y] comes from a bit string of the token starts kept in LDS behind the starts; with ~90 000 frames on
    a narrow workgroup there is no room for it and every frame bisects the starts instead.  Both roads, same answer."""
    rng = np.random.default_rng(5)
    for (B, Tx, Ty) in [(2, 3, 90000), (1, 40, 70001)]:
        v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        ty = np.array([Ty] + [int(rng.integers(Tx, Ty)) for _ in range(B - 1)], np.int32)
        tx = np.array([Tx] + [int(rng.integers(1, Tx + 1)) for _ in range(B - 1)], np.int32)
        want = _oracle_path(v, tx, ty)
        for kw in ({}, {"force_generic": True}):
            p, tok, dur = _hip(v, tx, ty, dev, **kw)
            assert np.array_equal(p, want), (B, Tx, Ty, kw)
            _check_consistency(p, tok, dur, tx, ty)





def test_two_workgroups_per_utterance(dev):
    """Text of 253..504 rows can run as two workgroups per utterance (two CUs; the boundary row between the halves
    travels through the workspace, the second half runs the backtrack).  The split must not change a bit: against the
    oracle, forced on at small sizes -- ragged batches where some utterances have no rows for the second half, long
    tokens, ties, a non-finite score in either half (exact fallback run by the second workgroup), 16-bit scores,
    strict masks -- and forced off."""
    rng = np.random.default_rng(2025)
    cases = [(3, 300, 700), (5, 504, 640), (2, 253, 2100), (4, 400, 1000), (9, 330, 512)]
    for it, (B, Tx, Ty) in enumerate(cases):
        v = (rng.standard_normal((B, Tx, Ty)) if it % 2 == 0 else rng.integers(-2, 3, (B, Tx, Ty))).astype(np.float32)
        for b in range(B):
            v[b, int(rng.integers(0, Tx)), :] += 2.5                         # a token that runs for hundreds of frames
        ty = rng.integers(Ty // 2, Ty + 1, B).astype(np.int32)
        ty[0] = Ty
        tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
        tx[0] = Tx
        if B > 1: tx[1] = min(200, ty[1])                                    # nothing for the second workgroup
        if B > 2: tx[2] = min(253, ty[2])                                    # one row for it
        want = _oracle_path(v, tx, ty)
        for cus in (2, 1):
            p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=cus)
            assert np.array_equal(p, want), (it, cus)
            _check_consistency(p, tok, dur, tx, ty)
    import aligner_amd
    assert aligner_amd.read_status(dev) == 0
    # non-finite scores: in the first half's rows, in the second's, in both
    B, Tx, Ty = 4, 420, 600
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    v[0, 17, 300] = np.inf
    v[1, 400, 450] = -np.inf
    v[2, 100, 200] = np.nan; v[2, 300, 500] = np.inf
    tx = np.array([420, 420, 420, 400], np.int32); ty = np.array([600, 590, 600, 600], np.int32)
    want = _oracle_path(v, tx, ty)
    for cus in (2, 1):
        p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=cus)
        assert np.array_equal(p, want), cus
    # a non-finite max_neg_val: no fast sweep at all, the second workgroup runs the exact one for the whole utterance
    B, Tx, Ty = 2, 300, 400
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([300, 260], np.int32); ty = np.array([400, 380], np.int32)
    want = _oracle_path(v, tx, ty, neg=-np.inf)
    for cus in (2, 1):
        p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=cus, max_neg_val=float("-inf"))
        assert np.array_equal(p, want), cus
    # 16-bit scores and a strict mask
    for dt in (torch.bfloat16, torch.float16):
        B, Tx, Ty = 3, 380, 960
        v16 = torch.from_numpy(rng.standard_normal((B, Tx, Ty)).astype(np.float32)).to(dt)
        tx = np.array([380, 300, 254], np.int32); ty = np.array([960, 900, 512], np.int32)
        want = _oracle_path(v16.float().numpy(), tx, ty)
        for cus in (2, 1):
            res = aligner_amd.align(v16.to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                                    path_dtype=torch.int32, cus_per_utterance=cus)
            assert np.array_equal(res.path.cpu().numpy(), want), (dt, cus)
    B, Tx, Ty = 2, 300, 640
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([300, 280], np.int32); ty = np.array([640, 600], np.int32)
    mask = np.zeros((B, Tx, Ty), np.float32)
    for b in range(B): mask[b, :tx[b], :ty[b]] = 1.0
    want = _oracle_path(v * mask, tx, ty)
    for cus in (2, 1):
        res = aligner_amd.align(torch.from_numpy(v).to(dev), mask=torch.from_numpy(mask).to(dev), strict_mask=True,
                                path_dtype=torch.int32, cus_per_utterance=cus)
        assert np.array_equal(res.path.cpu().numpy(), want), cus


def test_prezeroed_path_at_the_baseline_shapes(appendix_a, dev):
    """ALIGNER_F_PATH_PREZEROED at C2 (ragged), a C4 shard and C5 (two workgroups per utterance): the reference's hashes."""
    import aligner_amd
    rec, _ = appendix_a
    for tag in ("C2-varlen", "C4-shard0", "C5-longform"):
        v, tx, ty = _config(tag)
        out = torch.zeros(v.shape, dtype=torch.int32, device=dev)
        aligner_amd.align(torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                          out_path=out, out_path_is_zero=True)
        assert synth.sha256_of(out.cpu().numpy()) == rec[tag]["path_sha256"], tag
    assert aligner_amd.read_status(dev) == 0


def test_two_workgroups_each_half_walks_its_own_rows(dev):
    """Round 4: in the two-workgroup form each half walks its own rows out of its own LDS (hand-over of one frame number
    through the workspace) and stores its own part of the outputs.  Same bits as one walker for all rows (the debug option
    keeps that form) and as the oracle: ragged lengths, tokens hundreds of frames long across the hand-over, the hand-over
    landing on the first / last frame a row can own, every output dtype of the dense path, token index and durations."""
    import aligner_amd
    from aligner_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    cases = [(4, 500, 4000), (3, 300, 2100), (6, 504, 1900), (2, 253, 3000), (5, 420, 2048)]
    for it, (B, Tx, Ty) in enumerate(cases):
        v = (rng.standard_normal((B, Tx, Ty)) if it % 2 == 0 else rng.integers(-1, 2, (B, Tx, Ty))).astype(np.float32)
        ty = rng.integers(max(Tx, Ty // 2), Ty + 1, B).astype(np.int32)
        ty[0] = Ty
        tx = np.array([rng.integers(253, min(Tx, t) + 1) for t in ty], np.int32)
        tx[0] = Tx
        if B > 2: tx[2] = 253                                               # one row for the second half
        if B > 3: tx[3] = 200                                               # none: the first workgroup alone
        v[0, 251, :] += 3.0; v[0, 252, :] += 3.0                             # long tokens either side of the hand-over
        if B > 1:
            v[1, :252, :] -= 50.0                                           # the first half's rows as short as they can be:
            v[1, :252, :252] += 100.0 * np.eye(252, dtype=np.float32)       # the hand-over lands on frame 251
        want = _oracle_path(v, tx, ty)
        got = {}
        for no_split in (0, 1):
            try:
                assert lib.aligner_debug_set_option(b"maxpath_no_split_walk", no_split) == 0
                p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=2)
            finally:
                lib.aligner_debug_set_option(b"maxpath_no_split_walk", 0)
            assert np.array_equal(p, want), (it, no_split)
            _check_consistency(p, tok, dur, tx, ty)
            got[no_split] = (tok, dur)
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert aligner_amd.read_status(dev) == 0
    # the dense path in other dtypes (the zero workgroups ride in the same launch), 16-bit scores
    B, Tx, Ty = 3, 400, 2000
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([400, 333, 260], np.int32); ty = np.array([2000, 1999, 1800], np.int32)
    vb = torch.from_numpy(v).to(dev).to(torch.bfloat16)
    want = _oracle_path(vb.float().cpu().numpy(), tx, ty)
    for dt in (torch.float32, torch.uint8, torch.bfloat16, torch.int64):
        r = aligner_amd.align(vb, torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev), path_dtype=dt, cus_per_utterance=2)
        assert np.array_equal(r.path.to(torch.int32).cpu().numpy(), want), dt
        assert np.array_equal(r.durations.cpu().numpy(), want.sum(2))
    assert aligner_amd.read_status(dev) == 0


def test_two_workgroups_full_size_long_form(appendix_a, dev):
    """BASELINE config 5's shape through the form the library picks for it (two workgroups per utterance) and through
    the one-workgroup form: the same path, equal to the reference's hash."""
    import aligner_amd
    rec, _ = appendix_a
    v, tx, ty = _config("C5-longform")
    for cus in (None, 2, 1):
        p, _, dur = _hip(v, tx, ty, dev, cus_per_utterance=cus)
        assert synth.sha256_of(p) == rec["C5-longform"]["path_sha256"], cus
    assert aligner_amd.read_status(dev) == 0


def test_two_workgroups_in_a_replayed_graph(dev):
    """The two-workgroup form starts its launch by refilling the boundary ring (xring_fill_kernel on the caller's stream --
    a hipMemsetAsync node did not refill it on replay): captured into a HIP graph with the kernels, so a replay on NEW
    scores in the same buffers must not meet the
    previous replay's boundary rows.  C ABI straight, as a serving loop would drive it."""
    from aligner_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    B, Tx, Ty = 3, 330, 704
    tx = np.array([330, 300, 200], np.int32); ty = np.array([704, 650, 704], np.int32)
    d_v = torch.zeros((B, Tx, Ty), dtype=torch.float32, device=dev)
    d_tx, d_ty = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    path = torch.zeros((B, Tx, Ty), dtype=torch.int32, device=dev)
    dur = torch.zeros((B, Tx), dtype=torch.int32, device=dev)
    ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, Tx, Ty), dtype=torch.uint8, device=dev)

    def launch():
        _lib.check(lib.aligner_maxpath(d_v.data_ptr(), _lib.DT_F32, None, 0, d_tx.data_ptr(), d_ty.data_ptr(),
                                       path.data_ptr(), _lib.DT_I32, None, dur.data_ptr(), ws.data_ptr(), ws.numel(),
                                       B, Tx, Ty, -1e9, _lib.F_TWO_CUS, torch.cuda.current_stream().cuda_stream))

    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        launch()                                           # (first launch outside the capture: LDS attribute grants)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        launch()
    for trial in range(4):
        v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
        d_v.copy_(torch.from_numpy(v))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = _oracle_path(v, tx, ty)
        assert np.array_equal(path.cpu().numpy(), want), trial
        assert np.array_equal(dur.cpu().numpy(), want.sum(2)), trial
    st = np.zeros(1, np.int32)
    _lib.check(lib.aligner_maxpath_read_status(ws.data_ptr(), st.ctypes.data, None))
    assert int(st[0]) == 0


def test_two_workgroups_beside_other_work(dev):
    """The second workgroup of an utterance waits for the first through the workspace.  Run the form while other
    streams keep the CUs busy (a stream of alignment batches that occupy every CU, and a second two-workgroup launch
    on a third stream): the waits are bounded and either half may be scheduled late -- every result must still be the
    oracle's and the status word clean."""
    import aligner_amd
    rng = np.random.default_rng(77)
    B, Tx, Ty = 6, 420, 1800
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([420, 400, 300, 420, 260, 333], np.int32); ty = np.array([1800, 1700, 1800, 900, 1800, 1500], np.int32)
    want = _oracle_path(v, tx, ty)
    d_v, d_tx, d_ty = torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    # the load: [512,200,1000] batches, one workgroup per utterance on every CU, and a large elementwise kernel
    big = torch.from_numpy(synth.synth_value(512, 200, 1000, 3)).to(dev)
    btx = torch.full((512,), 200, dtype=torch.int32, device=dev); bty = torch.full((512,), 1000, dtype=torch.int32, device=dev)
    s_load, s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(s_load):
            for _ in range(4):
                aligner_amd.align(big, btx, bty, want_path=False)
                big2 = big * 1.0001
        with torch.cuda.stream(s_a):
            outs.append(aligner_amd.align(d_v, d_tx, d_ty, path_dtype=torch.int32, cus_per_utterance=2))
        with torch.cuda.stream(s_b):
            outs.append(aligner_amd.align(d_v, d_tx, d_ty, path_dtype=torch.int32, cus_per_utterance=2))
    torch.cuda.synchronize()
    for k, r in enumerate(outs):
        assert np.array_equal(r.path.cpu().numpy(), want), k
        assert np.array_equal(r.durations.cpu().numpy(), want.sum(2)), k
    assert aligner_amd.read_status(dev) == 0
    del big2


def test_two_workgroups_first_half_never_delivers(dev):
    """The defined failure of the two-workgroup form: if the first half does not deliver (here: switched off by the
    test flag), the second gives up after its bounded wait -- the utterance comes back all-zero with ALIGNER_ST_INTERNAL
    in the status word, align(check=True) raises, and maximum_path_c's host form returns an error.  Utterances that
    never needed a second workgroup are untouched."""
    import aligner_amd
    from aligner_amd import _lib
    rng = np.random.default_rng(5)
    B, Tx, Ty = 3, 300, 640
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([300, 200, 280], np.int32); ty = np.array([640, 600, 640], np.int32)     # utterance 1: one workgroup
    want = _oracle_path(v, tx, ty)
    assert aligner_amd.read_status(dev) == 0
    p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=2, _test_flags=_lib.F_TEST_DROP_FIRST_HALF)
    assert aligner_amd.read_status(dev) & _lib.ST_INTERNAL
    for b in (0, 2):
        assert not p[b].any() and not dur[b].any() and np.all(tok[b] == -1)
    assert np.array_equal(p[1], want[1]) and np.array_equal(dur[1], want[1].sum(1))
    with pytest.raises(RuntimeError, match="internal consistency"):
        aligner_amd.align(torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                          cus_per_utterance=2, _test_flags=_lib.F_TEST_DROP_FIRST_HALF, check=True)
    assert aligner_amd.read_status(dev) & _lib.ST_INTERNAL           # check=True does not clear the word
    # the blocking host form (maximum_path_c's contract) reports it as an error
    lib = _lib.load()
    paths = np.zeros((B, Tx, Ty), np.int32)
    vals = v.copy()
    rc = lib.aligner_maxpath_host_f32(paths.ctypes.data, vals.ctypes.data, tx.ctypes.data, ty.ctypes.data, B, Tx, Ty, -1e9,
                                      _lib.F_TWO_CUS | _lib.F_TEST_DROP_FIRST_HALF)
    assert rc == -5 and b"internal consistency" in lib.aligner_last_error()
    # and the next call is fine again
    p, _, _ = _hip(v, tx, ty, dev, cus_per_utterance=2)
    assert np.array_equal(p, want) and aligner_amd.read_status(dev) == 0
    # A first half that gives up waiting for the backtrack's hand-over (a contended chip; here: at once, by the test flag)
    # costs time, not the answer: the second half finds the cancelled word and walks every row itself.  The status word
    # still says that a wait gave up.
    p, tok, dur = _hip(v, tx, ty, dev, cus_per_utterance=2, _test_flags=_lib.F_TEST_IMPATIENT_FIRST_HALF)
    assert np.array_equal(p, want) and np.array_equal(dur, want.sum(2))
    st = aligner_amd.read_status(dev)
    assert st in (0, _lib.ST_INTERNAL)          # (0: the hand-over was there before the first poll)
    p, _, _ = _hip(v, tx, ty, dev, cus_per_utterance=2)
    assert np.array_equal(p, want) and aligner_amd.read_status(dev) == 0


def test_zero_workgroups_never_report_is_a_defined_failure(dev):
    """The defined failure of the one-launch dense path (zero workgroups beside the search): if the zero
    workgroups do not report (here: the test flag -- they write their zeros and keep quiet), every utterance's workgroup
    gives up after its bounded wait and writes NO one at all: the path comes back all zeros (never ones that a late zero
    workgroup might erase), durations zero, no token on any frame, ALIGNER_ST_INTERNAL raised; check=True raises, the
    blocking host form returns an error, and the next call is fine (the last workgroup resets both counters)."""
    import aligner_amd
    from aligner_amd import _lib
    rng = np.random.default_rng(11)
    B, Tx, Ty = 5, 70, 336                                   # B*Tx*Ty % 16 == 0: the zero workgroups store 16 bytes at a time
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx = np.array([70, 33, 1, 64, 70], np.int32); ty = np.array([336, 200, 9, 336, 70], np.int32)
    want = _oracle_path(v, tx, ty)
    assert aligner_amd.read_status(dev) == 0
    for dt in (torch.float32, torch.int32, torch.bfloat16, torch.uint8):
        vt, txt, tyt = torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
        r = aligner_amd.align(vt, txt, tyt, path_dtype=dt, want_tok=True, _test_flags=_lib.F_TEST_DROP_ZERO_REPORTS)
        torch.cuda.synchronize()
        assert aligner_amd.read_status(dev) & _lib.ST_INTERNAL, "the zero workgroups were not part of this launch?"
        assert not r.path.to(torch.float32).any() and not r.durations.any() and bool((r.tok == -1).all())
        r = aligner_amd.align(vt, txt, tyt, path_dtype=dt, want_tok=True)                # the same call without the fault
        assert np.array_equal(r.path.to(torch.int32).cpu().numpy(), want) and aligner_amd.read_status(dev) == 0
        assert np.array_equal(r.durations.cpu().numpy(), want.sum(2))
    with pytest.raises(RuntimeError, match="internal consistency"):
        aligner_amd.align(vt, txt, tyt, _test_flags=_lib.F_TEST_DROP_ZERO_REPORTS, check=True)
    assert aligner_amd.read_status(dev) & _lib.ST_INTERNAL
    lib = _lib.load()
    paths = np.zeros((B, Tx, Ty), np.int32)
    vals = v.copy()
    rc = lib.aligner_maxpath_host_f32(paths.ctypes.data, vals.ctypes.data, tx.ctypes.data, ty.ctypes.data, B, Tx, Ty, -1e9,
                                      _lib.F_TEST_DROP_ZERO_REPORTS)
    assert rc == -5 and b"internal consistency" in lib.aligner_last_error() and not paths.any()
    # the two-workgroup form carries zero workgroups too
    B2, Tx2, Ty2 = 2, 300, 1900
    v2 = rng.standard_normal((B2, Tx2, Ty2)).astype(np.float32)
    tx2 = np.array([300, 290], np.int32); ty2 = np.array([1900, 1800], np.int32)
    want2 = _oracle_path(v2, tx2, ty2)
    p, tok, dur = _hip(v2, tx2, ty2, dev, cus_per_utterance=2, _test_flags=_lib.F_TEST_DROP_ZERO_REPORTS)
    assert aligner_amd.read_status(dev) & _lib.ST_INTERNAL
    assert not p.any() and not dur.any() and np.all(tok == -1)
    p, tok, dur = _hip(v2, tx2, ty2, dev, cus_per_utterance=2)
    assert np.array_equal(p, want2) and aligner_amd.read_status(dev) == 0


def test_workspace_regrowth_keeps_the_status_word(dev):
    """A workspace that has to grow is replaced; the sticky status bits raised through the old one must survive
    until read_status() (StreamWorkspaces.get carries the word over on the stream)."""
    import aligner_amd
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        v = torch.zeros((2, 4, 8), device=dev)
        aligner_amd.align(v, torch.tensor([0, 3]), torch.tensor([8, 8]))               # t_x = 0: ALIGNER_ST_BAD_LENGTHS
        big = torch.zeros((4, 300, 2000), device=dev)
        aligner_amd.align(big, torch.full((4,), 300), torch.full((4,), 2000))          # same stream, larger workspace
    torch.cuda.synchronize()
    assert aligner_amd.read_status(dev) & 1
    assert aligner_amd.read_status(dev) == 0
    aligner_amd.maxpath.release_workspaces(dev, s.cuda_stream)
