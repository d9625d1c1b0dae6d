"""Out-of-bounds WRITE check of the HIP kernels (the reference compiles its loops with
boundscheck / wraparound off, core.pyx:7-8,38-39, and asks the rebuild to switch checks on in tests).

GPU AddressSanitizer is not available on this pool, so every buffer a kernel writes (dense path, token index,
durations, workspace, log-probs) is carved out of a larger allocation with canary bands on both sides and the
kernels are called through the C ABI on the interior pointers: a single stray store into a band fails the test.
Stray READS are not visible this way; the kernels' tile loaders use buffer resources sized to the utterance (reads
past it return 0), and the CPU restatement runs under -fsanitize=address,undefined (tests/test_oracle.py)."""
import numpy as np
import pytest
import torch

from aligner_amd import _lib, synth

pytestmark = pytest.mark.gpu

BAND = 4096            # bytes of canary on either side
CANARY = 0xA5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


class Fenced:
    """nbytes of device memory with a canary band before and after it."""

    def __init__(self, nbytes, dev, align=256):
        self.n = int(nbytes)
        pad = (-self.n) % align
        self.buf = torch.full((BAND + self.n + pad + BAND,), CANARY, dtype=torch.uint8, device=dev)
        self.lo, self.hi = BAND, BAND + self.n
        assert self.buf.data_ptr() % 256 == 0
        self.buf[self.lo:self.hi] = 0

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.lo

    def view(self, dtype, shape):
        return self.buf[self.lo:self.hi].view(dtype).reshape(shape)

    def intact(self):
        return bool((self.buf[:self.lo] == CANARY).all()) and bool((self.buf[self.hi:] == CANARY).all())


SHAPES = [(1, 1, 1), (3, 7, 13), (2, 63, 64), (4, 64, 257), (2, 127, 1000), (3, 200, 1001), (2, 253, 300),
          (1, 300, 700), (2, 505, 2100), (1, 600, 650), (3, 504, 1000), (2, 330, 3104)]


@pytest.mark.parametrize("B,Tx,Ty", SHAPES)
@pytest.mark.parametrize("flags", [0, _lib.F_FORCE_GENERIC, _lib.F_NO_PREV_TABLE, _lib.F_STREAM_PATH, _lib.F_TWO_CUS,
                                   _lib.F_SEPARATE_EXPAND])
def test_alignment_search_writes_stay_inside_their_buffers(dev, B, Tx, Ty, flags):
    lib = _lib.load()
    rng = np.random.default_rng(B * 1000 + Tx + Ty)
    value = torch.from_numpy(synth.synth_value(B, Tx, Ty, seed=Tx + Ty)).to(dev)
    ty = rng.integers(max(1, Ty // 2), Ty + 1, B).astype(np.int32)
    tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
    tx[0], ty[0] = Tx if Tx <= Ty else Ty, Ty
    d_tx, d_ty = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    wsb = lib.aligner_maxpath_workspace_bytes(B, Tx, Ty)
    ws, path = Fenced(wsb, dev), Fenced(B * Tx * Ty * 4, dev)
    tok, dur = Fenced(B * Ty * 4, dev), Fenced(B * Tx * 4, dev)
    _lib.check(lib.aligner_maxpath_f32(value.data_ptr(), None, 0, d_tx.data_ptr(), d_ty.data_ptr(), path.ptr, _lib.DT_I32,
                                       tok.ptr, dur.ptr, ws.ptr, wsb, B, Tx, Ty, -1e9, flags,
                                       torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for name, f in (("workspace", ws), ("path", path), ("tok", tok), ("durations", dur)):
        assert f.intact(), f"{name}: a kernel wrote outside its buffer"
    # and the result is the right one (the fences did not move anything)
    from oracle import maxpath_oracle as O
    want = np.zeros((B, Tx, Ty), np.int32)
    O.maximum_path_c(want, value.cpu().numpy().copy(), tx, ty, -1e9)
    assert np.array_equal(path.view(torch.int32, (B, Tx, Ty)).cpu().numpy(), want)
    assert np.array_equal(dur.view(torch.int32, (B, Tx)).cpu().numpy(), want.sum(2))


DT_TORCH = {_lib.DT_F32: torch.float32, _lib.DT_F16: torch.float16, _lib.DT_BF16: torch.bfloat16, _lib.DT_F64: torch.float64,
            _lib.DT_I32: torch.int32, _lib.DT_U8: torch.uint8, _lib.DT_I64: torch.int64}


@pytest.mark.parametrize("B,Tx,Ty", [(1, 1, 1), (3, 7, 13), (2, 63, 65), (3, 200, 1001), (2, 330, 704)])
@pytest.mark.parametrize("pdt", list(DT_TORCH))
@pytest.mark.parametrize("misalign", [0, 2])
def test_zero_and_scatter_writes_stay_inside_their_buffers(dev, B, Tx, Ty, pdt, misalign):
    """The dense path in two steps (zeros beside the search, then one 1 per frame): every dtype, odd sizes, and a path
    pointer that is not 16-byte aligned (the zero kernel's byte-wise head and tail); equal to expand, inside the fence."""
    lib = _lib.load()
    es = torch.empty((), dtype=DT_TORCH[pdt]).element_size()
    off = misalign * es
    rng = np.random.default_rng(B + Tx + Ty)
    value = torch.from_numpy(synth.synth_value(B, Tx, Ty, seed=Tx * Ty)).to(dev)
    ty = rng.integers(max(1, Ty // 2), Ty + 1, B).astype(np.int32)
    tx = np.array([rng.integers(1, min(Tx, t) + 1) for t in ty], np.int32)
    tx[0], ty[0] = min(Tx, Ty), Ty
    d_tx, d_ty = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    wsb = lib.aligner_maxpath_workspace_bytes(B, Tx, Ty)
    ws = Fenced(wsb, dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.aligner_maxpath_forward_f32(value.data_ptr(), None, 0, d_tx.data_ptr(), d_ty.data_ptr(), None, None, ws.ptr,
                                               wsb, B, Tx, Ty, -1e9, 0, s))
    n = B * Tx * Ty * es
    two = Fenced(n + off, dev)
    two.buf[two.lo:two.hi] = 0x77                                  # stale contents the zeros must overwrite
    two.buf[two.lo:two.lo + off] = CANARY                          # (the bytes before a misaligned path are fence too)
    for fl in (0, _lib.F_STREAM_PATH):
        _lib.check(lib.aligner_maxpath_zero_path(two.ptr + off, pdt, B, Tx, Ty, fl, s))
        _lib.check(lib.aligner_maxpath_scatter_path(ws.ptr, two.ptr + off, pdt, B, Tx, Ty, s))
        torch.cuda.synchronize()
        assert two.intact() and bool((two.buf[two.lo:two.lo + off] == CANARY).all()), "zero/scatter wrote outside the path"
        got = two.buf[two.lo + off:two.hi].clone().view(DT_TORCH[pdt]).reshape(B, Tx, Ty)
        if off == 0:
            one = Fenced(n, dev)
            _lib.check(lib.aligner_maxpath_expand(ws.ptr, one.ptr, pdt, B, Tx, Ty, s))
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.uint8), one.view(DT_TORCH[pdt], (B, Tx, Ty)).view(torch.uint8))
        from oracle import maxpath_oracle as O
        want = np.zeros((B, Tx, Ty), np.int32)
        O.maximum_path_c(want, value.cpu().numpy().copy(), tx, ty, -1e9)
        assert np.array_equal(got.to(torch.int32).cpu().numpy(), want)
        two.buf[two.lo + off:two.hi] = 0x77
    assert ws.intact()


@pytest.mark.parametrize("B,C,Tx,Ty", [(2, 80, 50, 130), (3, 80, 200, 1000), (1, 16, 33, 65), (2, 80, 257, 300),
                                       (1, 80, 500, 1030),
                                       # the row-tile form (frames in quads, C = 16 KS): a partial strip, one frame quad, KS = 8
                                       (3, 80, 199, 36), (2, 80, 224, 4), (2, 128, 100, 260)])
@pytest.mark.parametrize("logp_dt", [_lib.DT_F32, _lib.DT_BF16])
def test_similarity_writes_stay_inside_their_buffers(dev, B, C, Tx, Ty, logp_dt):
    lib = _lib.load()
    g = torch.Generator().manual_seed(C + Tx + Ty)
    k = torch.randn(B, C, Tx, generator=g).to(dev)
    q = torch.randn(B, C, Ty, generator=g).to(dev)
    t_x = torch.tensor([Tx] + [max(1, Tx - 7 * i) for i in range(1, B)], dtype=torch.int32, device=dev)
    wsb = lib.aligner_softattn_workspace_bytes(B, C, Tx)
    ws = Fenced(max(wsb, 256), dev)
    esz = 4 if logp_dt == _lib.DT_F32 else 2
    logp = Fenced(B * Tx * Ty * esz, dev)
    _lib.check(lib.aligner_softattn(k.data_ptr(), q.data_ptr(), t_x.data_ptr(), None, logp.ptr, logp_dt, None, ws.ptr, wsb,
                                    B, C, Tx, Ty, 0.0005, _lib.SIM_L2, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert logp.intact(), "softattn wrote outside the log-prob buffer"
    assert ws.intact(), "softattn wrote outside its workspace"
    out = logp.view(torch.float32 if esz == 4 else torch.bfloat16, (B, Tx, Ty)).float()
    assert bool(torch.isfinite(out[0]).all())            # utterance 0 uses every text row: no masked (-inf) row
    col = torch.logsumexp(out[0], dim=0)
    assert float(col.abs().max()) < (1e-3 if esz == 4 else 5e-2)


@pytest.mark.parametrize("B,Tx,Ty", [(2, 9, 40), (3, 63, 200), (2, 200, 1000), (2, 260, 400), (1, 505, 1100), (1, 700, 900)])
def test_forward_sum_prior_regulator_writes_stay_inside_their_buffers(dev, B, Tx, Ty):
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tx * 7 + Ty)
    logp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    t_x = torch.tensor([Tx] + [max(1, Tx - 5 * i) for i in range(1, B)], dtype=torch.int32, device=dev)
    t_y = torch.tensor([Ty] + [max(Tx, Ty - 11 * i) for i in range(1, B)], dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    wsb = lib.aligner_forward_sum_workspace_bytes(B, Tx, Ty)
    ws, loss, grad = Fenced(wsb, dev), Fenced(B * 4, dev), Fenced(B * Tx * Ty * 4, dev)
    _lib.check(lib.aligner_forward_sum_f32(logp.data_ptr(), t_x.data_ptr(), t_y.data_ptr(), loss.ptr, grad.ptr, ws.ptr, wsb,
                                           B, Tx, Ty, s))
    prior = Fenced(B * Tx * Ty * 4, dev)
    _lib.check(lib.aligner_beta_binomial_prior_f32(t_x.data_ptr(), t_y.data_ptr(), prior.ptr, B, Tx, Ty, 1.0, s))
    C = 24
    h = torch.randn(B, C, Tx, generator=g).to(dev)
    dur = torch.zeros((B, Tx), dtype=torch.int32, device=dev)
    dur[:, 0] = t_y - (t_x - 1)
    for b in range(B):
        dur[b, 1:int(t_x[b])] = 1
    out, tok = Fenced(B * C * Ty * 4, dev), Fenced(B * Ty * 4, dev)
    _lib.check(lib.aligner_regulate_f32(h.data_ptr(), dur.data_ptr(), out.ptr, tok.ptr, B, C, Tx, Ty, s))
    torch.cuda.synchronize()
    for name, f in (("forward-sum workspace", ws), ("loss", loss), ("grad", grad), ("prior", prior), ("regulated", out),
                    ("regulator tok", tok)):
        assert f.intact(), f"{name}: a kernel wrote outside its buffer"
    assert bool(torch.isfinite(loss.view(torch.float32, (B,))).all())
    post = -grad.view(torch.float32, (B, Tx, Ty))[0, :, :int(t_y[0])].sum(0)       # a frame's posterior sums to 1
    assert float((post - 1).abs().max()) < 2e-3


@pytest.mark.parametrize("B,Tx,Ty,D", [(2, 5, 9, 3), (2, 40, 300, 16), (1, 100, 600, 7), (2, 64, 257, 32), (3, 33, 700, 40),
                                       (1, 6, 2600, 1300), (1, 120, 1000, 16)])
@pytest.mark.parametrize("exact_ws", [False, True])
@pytest.mark.parametrize("want_la", [True, False])
def test_boundary_search_writes_stay_inside_their_buffers(dev, B, Tx, Ty, D, exact_ws, want_la):
    """Normalisers, the segmented chain (ring, trash area, per-(token, position) durations), backtrack and gamma kernels:
    ragged lengths, one and many segments per utterance, several positions per thread; with the workspace of the
    window-independent bound and with exactly aligner_boundary_search_workspace_bytes_ex bytes."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tx + Ty + D)
    e = torch.randn(B, Tx, Ty, generator=g).to(dev)
    tx = [Tx] + [max(-(-Ty // (2 * D)), Tx - 3 * i) for i in range(1, B)]
    ty = [min(Ty, Tx * D)] + [min(Ty - 7 * i, tx[i] * D) for i in range(1, B)]
    t_x = torch.tensor(tx, dtype=torch.int32, device=dev)
    t_y = torch.tensor(ty, dtype=torch.int32, device=dev)
    wsb = lib.aligner_boundary_search_workspace_bytes_ex(B, Tx, Ty, D) if exact_ws else lib.aligner_boundary_search_workspace_bytes(B, Tx, Ty)
    assert 0 < wsb and lib.aligner_boundary_search_workspace_bytes_ex(B, Tx, Ty, D) <= lib.aligner_boundary_search_workspace_bytes(B, Tx, Ty)
    ws, bnd, dur, sc = Fenced(wsb, dev), Fenced(B * Tx * 4, dev), Fenced(B * Tx * 4, dev), Fenced(B * 4, dev)
    la, gm = Fenced(B * Tx * Ty * 4, dev), Fenced(B * Tx * Ty * 4, dev)
    for _ in range(2):                                   # twice: the second call meets the first one's ring and counters
        # (without log_alpha the chain runs as the max-product kernel alone)
        _lib.check(lib.aligner_boundary_search(e.data_ptr(), _lib.DT_F32, t_x.data_ptr(), t_y.data_ptr(), D, bnd.ptr, dur.ptr,
                                               sc.ptr, la.ptr if want_la else None, gm.ptr if want_la else None, ws.ptr, wsb,
                                               B, Tx, Ty, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for name, f in (("workspace", ws), ("boundaries", bnd), ("durations", dur), ("score", sc), ("log_alpha", la), ("gamma", gm)):
        assert f.intact(), f"{name}: a kernel wrote outside its buffer"
    d = dur.view(torch.int32, (B, Tx))
    for b in range(B):
        db = d[b, :tx[b]]
        assert bool((db >= 1).all()) and bool((db <= D).all()) and int(db.sum()) == ty[b]
    assert int(ws.view(torch.int32, (wsb // 4,))[0]) == 0          # status word


@pytest.mark.parametrize("B,Tx,Ty,D", [(2, 5, 9, 3), (2, 40, 300, 16), (1, 100, 600, 7), (2, 64, 257, 32), (3, 33, 700, 40),
                                       (1, 6, 2600, 1300), (1, 120, 1000, 16), (1, 10, 1001, 250)])
@pytest.mark.parametrize("general", [0, 1])
def test_boundary_search_gradient_writes_stay_inside_their_buffers(dev, B, Tx, Ty, D, general):
    """The gradient's kernels (normalisers, cotangent, chain in its split and its general form, per-cell gradient) with
    exactly aligner_boundary_search_backward_workspace_bytes bytes; twice, and the two forms of the chain agree."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tx + Ty + D)
    e = torch.randn(B, Tx, Ty, generator=g).to(dev)
    w1, w2 = torch.randn(B, Tx, Ty, generator=g).to(dev), torch.randn(B, Tx, Ty, generator=g).to(dev)
    tx = [Tx] + [max(-(-Ty // (2 * D)), Tx - 3 * i) for i in range(1, B)]
    ty = [min(Ty, Tx * D)] + [min(Ty - 7 * i, tx[i] * D) for i in range(1, B)]
    t_x = torch.tensor(tx, dtype=torch.int32, device=dev)
    t_y = torch.tensor(ty, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    wsf = lib.aligner_boundary_search_workspace_bytes_ex(B, Tx, Ty, D)
    wf = torch.zeros(wsf, dtype=torch.uint8, device=dev)
    bnd = torch.empty(B, Tx, dtype=torch.int32, device=dev)
    la = torch.empty(B, Tx, Ty, dtype=torch.float32, device=dev)
    _lib.check(lib.aligner_boundary_search(e.data_ptr(), _lib.DT_F32, t_x.data_ptr(), t_y.data_ptr(), D, bnd.data_ptr(), None,
                                           None, la.data_ptr(), None, wf.data_ptr(), wsf, B, Tx, Ty, stream))
    wsb = lib.aligner_boundary_search_backward_workspace_bytes(B, Tx, Ty, D)
    assert wsb > 0
    ws, grad = Fenced(wsb, dev), Fenced(B * Tx * Ty * 4, dev)
    outs = []
    try:
        for form in (general, general, 1 - general):
            assert lib.aligner_debug_set_option(b"mobo_bwd_general", form) == 0
            _lib.check(lib.aligner_boundary_search_backward(e.data_ptr(), _lib.DT_F32, t_x.data_ptr(), t_y.data_ptr(), D,
                                                            la.data_ptr(), w1.data_ptr(), w2.data_ptr(), grad.ptr, ws.ptr, wsb,
                                                            B, Tx, Ty, stream))
            torch.cuda.synchronize()
            outs.append(grad.view(torch.float32, (B, Tx, Ty)).clone())
    finally:
        lib.aligner_debug_set_option(b"mobo_bwd_general", 0)
    for name, f in (("workspace", ws), ("gradient", grad)):
        assert f.intact(), f"{name}: a kernel wrote outside its buffer"
    assert torch.equal(outs[0], outs[1])
    scale = float(outs[0].abs().max())
    assert scale > 0 and float((outs[0] - outs[2]).abs().max()) < 1e-4 * scale
    assert int(ws.view(torch.int32, (wsb // 4,))[0]) == 0          # status word
    too_small = lib.aligner_boundary_search_backward(e.data_ptr(), _lib.DT_F32, t_x.data_ptr(), t_y.data_ptr(), D, la.data_ptr(),
                                                     w1.data_ptr(), None, grad.ptr, ws.ptr, wsb - 1, B, Tx, Ty, stream)
    assert too_small == -28                                         # ALIGNER_ENOSPC


@pytest.mark.parametrize("B,Tx,Ty", [(2, 200, 1000), (3, 63, 200), (2, 300, 504), (129, 30, 64)])
def test_forward_sum_serial_form_writes_stay_inside_their_buffers(dev, request, B, Tx, Ty):
    """forward, then the gradient-making backward kernel with its hand-issued stager (fs_grad_stager_by_hand): the form
    batches past half the CU count take; forced on the small ones."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tx * 3 + Ty)
    logp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    t_x = torch.tensor([Tx] + [max(1, Tx - 5 * (i % 7)) for i in range(1, B)], dtype=torch.int32, device=dev)
    t_y = torch.tensor([Ty] + [max(Tx, Ty - 3 * (i % 11)) for i in range(1, B)], dtype=torch.int32, device=dev)
    wsb = lib.aligner_forward_sum_workspace_bytes(B, Tx, Ty)
    ws, loss, grad = Fenced(wsb, dev), Fenced(B * 4, dev), Fenced(B * Tx * Ty * 4, dev)
    request.addfinalizer(lambda: lib.aligner_debug_set_option(b"fwdsum_serial", 0))
    _lib.check(lib.aligner_debug_set_option(b"fwdsum_serial", 1))
    _lib.check(lib.aligner_forward_sum_f32(logp.data_ptr(), t_x.data_ptr(), t_y.data_ptr(), loss.ptr, grad.ptr, ws.ptr, wsb,
                                           B, Tx, Ty, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for name, f in (("forward-sum workspace", ws), ("loss", loss), ("grad", grad)):
        assert f.intact(), f"{name}: a kernel wrote outside its buffer"
    post = -grad.view(torch.float32, (B, Tx, Ty))[0, :, :int(t_y[0])].sum(0)       # a frame's posterior sums to 1
    assert float((post - 1).abs().max()) < 1e-3


@pytest.mark.parametrize("B,Tx,Ty", [(2, 9, 40), (2, 200, 1000), (2, 255, 400), (1, 300, 500), (1, 600, 900)])
def test_forward_sum_ctc_form_writes_stay_inside_their_buffers(dev, B, Tx, Ty):
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tx * 5 + Ty)
    x = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    t_x = torch.tensor([Tx] + [max(1, Tx - 5 * i) for i in range(1, B)], dtype=torch.int32, device=dev)
    t_y = torch.tensor([Ty] + [max(Tx, Ty - 11 * i) for i in range(1, B)], dtype=torch.int32, device=dev)
    wsb = lib.aligner_forward_sum_ctc_workspace_bytes(B, Tx, Ty)
    ws, loss, grad = Fenced(wsb, dev), Fenced(B * 4, dev), Fenced(B * Tx * Ty * 4, dev)
    _lib.check(lib.aligner_forward_sum_ctc_f32(x.data_ptr(), t_x.data_ptr(), t_y.data_ptr(), -1.0, loss.ptr, grad.ptr, ws.ptr, wsb,
                                               B, Tx, Ty, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for name, f in (("workspace", ws), ("loss", loss), ("grad", grad)):
        assert f.intact(), f"{name}: a CTC-form kernel wrote outside its buffer"
    assert bool(torch.isfinite(loss.view(torch.float32, (B,))).all())


@pytest.mark.parametrize("B,Tx,Ty", [(3, 7, 13), (2, 127, 1000), (2, 505, 700)])
def test_running_scores_written_in_place_stay_inside_the_score_block(dev, B, Tx, Ty):
    """ALIGNER_F_WRITE_Q on device pointers: the score block itself is written (core.pyx:30), and nothing around it."""
    lib = _lib.load()
    from oracle import maxpath_oracle as O
    rng = np.random.default_rng(Tx + Ty)
    v = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    ty = rng.integers(max(Tx, Ty // 2), Ty + 1, B).astype(np.int32)
    tx = np.array([rng.integers(1, Tx + 1) for _ in ty], np.int32)
    tx[0], ty[0] = Tx, Ty
    val = Fenced(B * Tx * Ty * 4, dev)
    val.view(torch.float32, (B, Tx, Ty)).copy_(torch.from_numpy(v))
    d_tx, d_ty = torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)
    wsb = lib.aligner_maxpath_workspace_bytes(B, Tx, Ty)
    ws, path = Fenced(wsb, dev), Fenced(B * Tx * Ty * 4, dev)
    _lib.check(lib.aligner_maxpath_f32(val.ptr, None, 0, d_tx.data_ptr(), d_ty.data_ptr(), path.ptr, _lib.DT_I32, None, None,
                                       ws.ptr, wsb, B, Tx, Ty, -1e9, _lib.F_WRITE_Q, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert val.intact() and ws.intact() and path.intact()
    want_p, want_q = np.zeros(v.shape, np.int32), v.copy()
    O.maximum_path_c(want_p, want_q, tx, ty, -1e9)
    assert np.array_equal(path.view(torch.int32, (B, Tx, Ty)).cpu().numpy(), want_p)
    assert np.array_equal(val.view(torch.float32, (B, Tx, Ty)).cpu().numpy().view(np.uint32), want_q.view(np.uint32))
