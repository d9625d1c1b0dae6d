"""Gradient of the MoBoAligner boundary search ("log-domain alpha / beta").

PARITY UNPINNED like the search itself (no MoBoAligner source in the snapshot).  The checker is
oracle/mobo_oracle.py::boundary_search_backward -- the analytic adjoint of the alpha recursion in float64 -- which is
pinned here to (i) autograd over an independent float64 torch restatement of the forward pass and (ii) central
differences of the plain-loop forward oracle.  The GPU tests compare the HIP path with it through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import mobo_oracle as M

gpu = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------------
# the oracle's own pins (CPU)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("I,J,D", [(3, 7, 4), (5, 20, 6), (8, 40, 9), (4, 16, 4), (1, 5, 8), (6, 6, 3), (10, 100, 16),
                                   (7, 30, 30)])
def test_oracle_adjoint_equals_autograd_of_the_torch_restatement(I, J, D):
    rng = np.random.default_rng(I * 1000 + J)
    e = rng.normal(size=(I, J)) * 2
    ref = M.boundary_search_fast(e, D)
    et = torch.tensor(e, requires_grad=True)
    la, ga, alive = M.boundary_search_torch(et, D)
    fin = np.isfinite(ref["log_alpha"])
    assert np.array_equal(alive.numpy(), fin)
    assert np.allclose(ref["log_alpha"][fin], la.detach().numpy()[fin], atol=1e-10)
    assert np.allclose(ref["gamma"], ga.detach().numpy(), atol=1e-10)
    g1, g2 = rng.normal(size=(I, J)), rng.normal(size=(I, J))
    loss = (torch.where(alive, la, torch.zeros_like(la)) * torch.tensor(g1)).sum() + (ga * torch.tensor(g2)).sum()
    loss.backward()
    got = M.boundary_search_backward(e, D, g1, g2)
    assert np.abs(got - et.grad.numpy()).max() < 1e-11
    assert np.abs(got.sum(axis=1)).max() < 1e-11                 # a token row's energies only count up to a shift
    both = M.boundary_search_backward(e, D, g1, None) + M.boundary_search_backward(e, D, None, g2)
    assert np.abs(both - got).max() < 1e-11


def test_oracle_adjoint_equals_central_differences_of_the_loop_oracle():
    rng = np.random.default_rng(5)
    I, J, D = 4, 11, 5
    e = rng.normal(size=(I, J))
    g1, g2 = rng.normal(size=(I, J)), rng.normal(size=(I, J))

    def loss(x):
        r = M.boundary_search(x, D)
        la = np.where(np.isfinite(r["log_alpha"]), r["log_alpha"], 0.0)
        return (la * g1).sum() + (r["gamma"] * g2).sum()

    got = M.boundary_search_backward(e, D, g1, g2)
    h = 1e-6
    for i in range(I):
        for y in range(J):
            d = np.zeros_like(e)
            d[i, y] = h
            fd = (loss(e + d) - loss(e - d)) / (2 * h)
            assert abs(fd - got[i, y]) < 1e-7, (i, y, fd, got[i, y])


def test_oracle_adjoint_with_masked_energies():
    """-inf energies: no mass, no gradient there, and the rest is the gradient of the remaining chain."""
    rng = np.random.default_rng(6)
    I, J, D = 5, 24, 7
    e = rng.normal(size=(I, J))
    mask = rng.random((I, J)) < 0.15
    mask[:, -1] = False
    e_m = np.where(mask, -np.inf, e)
    r = M.boundary_search_fast(e_m, D)
    assert np.isfinite(r["map_score"])
    g1 = rng.normal(size=(I, J))
    got = M.boundary_search_backward(e_m, D, g1, None)
    assert np.all(got[mask] == 0) and np.all(np.isfinite(got))
    big = M.boundary_search_backward(np.where(mask, -700.0, e), D, np.where(np.isfinite(r["log_alpha"]), g1, 0.0), None)
    assert np.abs(np.where(mask, 0.0, big) - got).max() < 1e-9


# ------------------------------------------------------------------------------------------------------------
# the HIP path
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _lengths(rng, B, Tx, Ty, D):
    tx = np.array([Tx] + [int(rng.integers(max(1, -(-Ty // (2 * D))), Tx + 1)) for _ in range(B - 1)], np.int32)
    ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, B)], np.int32)
    if Ty > Tx * D:
        ty[0] = Tx * D
    return tx, ty


def _check_grad(dev, e, tx, ty, D, use_la=True, use_ga=True, dt=torch.float32, seed=0, rtol=3e-3, only=None):
    """HIP gradient against the float64 adjoint; tolerance relative to the utterance's largest gradient entry (the HIP
    path starts from its own fp32 log_alpha, whose absolute error grows with the depth of the chain); cotangents are
    standard normal."""
    import aligner_amd
    rng = np.random.default_rng(seed)
    B, Tx, Ty = e.shape
    ed = torch.from_numpy(e).to(dt).to(dev)
    txd, tyd = torch.from_numpy(tx), torch.from_numpy(ty)
    r = aligner_amd.boundary_search(ed, txd, tyd, D, want_log_alpha=True)
    g1 = rng.standard_normal((B, Tx, Ty)).astype(np.float32) if use_la else None
    g2 = rng.standard_normal((B, Tx, Ty)).astype(np.float32) if use_ga else None
    grad = aligner_amd.boundary_search_backward(
        ed, txd, tyd, D, r.log_alpha, None if g1 is None else torch.from_numpy(g1).to(dev),
        None if g2 is None else torch.from_numpy(g2).to(dev))
    torch.cuda.synchronize()
    got = grad.cpu().numpy().astype(np.float64)
    e64 = ed.float().cpu().numpy().astype(np.float64)
    worst = 0.0
    for b in (range(B) if only is None else only):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_backward(e64[b, :I, :J], D, None if g1 is None else g1[b, :I, :J].astype(np.float64),
                                          None if g2 is None else g2[b, :I, :J].astype(np.float64))
        # relative to the utterance's largest entry, plus the rounding of Z - flow (two O(|cotangent|) terms) itself
        err = np.abs(got[b, :I, :J] - want).max()
        tol = rtol * (1 + I / 100) * np.abs(want).max() + 2e-5
        worst = max(worst, err / tol)
        assert err < tol, (b, err, tol)
        assert np.all(got[b, I:] == 0) and np.all(got[b, :, J:] == 0), b
    return worst


@gpu
@pytest.mark.parametrize("B,Tx,Ty,D", [(3, 1, 1, 1), (2, 5, 9, 3), (4, 12, 40, 8), (3, 40, 300, 16), (2, 64, 257, 32),
                                       (2, 30, 1100, 64), (1, 100, 600, 7), (2, 9, 90, 10),
                                       (1, 120, 1000, 16),      # one utterance over 16 position segments
                                       (5, 33, 700, 40),        # ragged utterances: fewer segments than the launch has
                                       (2, 8, 1500, 800),       # a window of half the utterance: one segment, 1501 positions
                                       (1, 6, 2600, 1300),      # ... several positions per thread (state in LDS)
                                       (1, 10, 1001, 250),      # a last segment shorter than the window
                                       (300, 4, 20, 6)])        # more utterances than CUs
def test_gradient_matches_oracle(dev, B, Tx, Ty, D):
    rng = np.random.default_rng(B * 100 + Tx)
    e = (rng.standard_normal((B, Tx, Ty)) * 2).astype(np.float32)
    tx, ty = _lengths(rng, B, Tx, Ty, D)
    _check_grad(dev, e, tx, ty, D, seed=Tx, only=None if B <= 8 else range(0, B, 37))


@gpu
@pytest.mark.parametrize("use_la,use_ga", [(True, False), (False, True)])
def test_gradient_of_either_cotangent_alone(dev, use_la, use_ga):
    rng = np.random.default_rng(12)
    e = (rng.standard_normal((3, 25, 400)) * 2).astype(np.float32)
    tx, ty = np.array([25, 20, 14], np.int32), np.array([400, 333, 160], np.int32)
    _check_grad(dev, e, tx, ty, 24, use_la=use_la, use_ga=use_ga)


@gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gradient_sixteen_bit_energies(dev, dt):
    rng = np.random.default_rng(8)
    e = (rng.standard_normal((2, 20, 150)) * 2).astype(np.float32)
    _check_grad(dev, e, np.array([20, 13], np.int32), np.array([150, 90], np.int32), 16, dt=dt)


@gpu
def test_gradient_with_energies_of_sixty_nats(dev):
    """Deep tails: u and q of neighbouring positions differ by hundreds of nats; every window term is its own
    exponential of a non-positive sum, so nothing over- or underflows on the way."""
    rng = np.random.default_rng(21)
    e = (rng.standard_normal((2, 40, 500)) * 30).astype(np.float32)
    _check_grad(dev, e, np.array([40, 31], np.int32), np.array([500, 420], np.int32), 24, rtol=1e-2)


@gpu
def test_gradient_masked_and_infeasible_utterances(dev):
    import aligner_amd
    rng = np.random.default_rng(33)
    B, Tx, Ty, D = 3, 12, 80, 10
    e = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    mask = rng.random((Tx, Ty)) < 0.1
    mask[:, -1] = False
    e[0][mask] = -np.inf
    tx, ty = np.array([12, 3, 12], np.int32), np.array([80, 80, 70], np.int32)     # utterance 1: 3 tokens x 10 < 80 frames
    ed = torch.from_numpy(e).to(dev)
    r = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D, want_log_alpha=True)
    g = torch.randn(B, Tx, Ty, generator=torch.Generator().manual_seed(1)).to(dev)
    grad = aligner_amd.boundary_search_backward(ed, torch.from_numpy(tx), torch.from_numpy(ty), D, r.log_alpha, g, g)
    torch.cuda.synchronize()
    got = grad.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got)) and np.all(got[1] == 0) and np.all(got[0][mask] == 0)
    gn = g.cpu().numpy().astype(np.float64)
    for b in (0, 2):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_backward(e[b, :I, :J].astype(np.float64), D, gn[b, :I, :J], gn[b, :I, :J])
        assert np.abs(got[b, :I, :J] - want).max() < 3e-3 * max(np.abs(want).max(), 1e-3), b


@gpu
def test_soft_boundaries_backpropagates_through_autograd(dev):
    import aligner_amd
    rng = np.random.default_rng(40)
    B, Tx, Ty, D = 2, 16, 200, 20
    e = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    tx, ty = np.array([16, 11], np.int32), np.array([200, 150], np.int32)
    w = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    ed = torch.from_numpy(e).to(dev).requires_grad_(True)
    r = aligner_amd.soft_boundaries(ed, torch.from_numpy(tx), torch.from_numpy(ty), D)
    assert r.gamma.requires_grad and r.log_alpha.requires_grad and not r.durations.requires_grad
    (r.gamma * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = ed.grad.cpu().numpy().astype(np.float64)
    for b in range(B):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_backward(e[b, :I, :J].astype(np.float64), D, None, w[b, :I, :J].astype(np.float64))
        assert np.abs(got[b, :I, :J] - want).max() < 3e-3 * np.abs(want).max(), b
    plain = aligner_amd.boundary_search(ed.detach(), torch.from_numpy(tx), torch.from_numpy(ty), D, want_gamma=True)
    assert torch.equal(plain.gamma, r.gamma.detach()) and torch.equal(plain.boundaries, r.boundaries)
    # the training form: no MAP outputs, the forward pass on the sum-product chain alone -- the same gradient
    ed2 = torch.from_numpy(e).to(dev).requires_grad_(True)
    r2 = aligner_amd.soft_boundaries(ed2, torch.from_numpy(tx), torch.from_numpy(ty), D, want_map=False)
    assert r2.boundaries is None and torch.equal(r2.gamma.detach(), r.gamma.detach())
    (r2.gamma * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(ed2.grad, ed.grad)


@gpu
def test_gradient_at_config5_size(dev):
    """[8,500,4000], max duration 32: two utterances against the float64 adjoint (seconds each), the others through
    the property every token row has -- its gradient sums to zero (its energies only count up to a shift)."""
    g = torch.Generator().manual_seed(2)
    e = (torch.randn(8, 500, 4000, generator=g) * 2).numpy()
    rng = np.random.default_rng(50)
    tx = np.array([500, 500, 480, 450, 400, 333, 250, 130], np.int32)
    ty = np.array([4000, 3900, 4000, 3600, 3200, 2700, 3000, 4000], np.int32)
    import aligner_amd
    ed = torch.from_numpy(e).to(dev)
    r = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), 32, want_log_alpha=True)
    g2 = torch.randn(8, 500, 4000, generator=g).to(dev)
    grad = aligner_amd.boundary_search_backward(ed, torch.from_numpy(tx), torch.from_numpy(ty), 32, r.log_alpha, None, g2)
    torch.cuda.synchronize()
    got = grad.double().cpu().numpy()
    assert np.all(np.isfinite(got))
    scale = np.abs(got).max(axis=(1, 2))
    assert np.all(scale > 1e-2)
    assert np.all(np.abs(got.sum(axis=2)).max(axis=1) < 2e-3 * scale)
    g2n = g2.double().cpu().numpy()
    for b in (1, 7):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_backward(e[b, :I, :J].astype(np.float64), 32, None, g2n[b, :I, :J])
        assert np.abs(got[b, :I, :J] - want).max() < 2e-2 * np.abs(want).max(), b
        assert np.all(got[b, I:] == 0) and np.all(got[b, :, J:] == 0)


@gpu
def test_gradient_segment_that_never_delivers_fails_loudly(dev):
    import aligner_amd
    from aligner_amd import _lib, mobo
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    e = torch.randn(2, 50, 900, generator=g).to(dev)
    tx, ty = torch.tensor([50, 40]), torch.tensor([900, 700])
    torch.cuda.synchronize()
    mobo.read_status(dev)                                   # (clears what earlier tests left)
    r = aligner_amd.boundary_search(e, tx, ty, 32, want_log_alpha=True)
    w = torch.randn(2, 50, 900, generator=g).to(dev)
    good = aligner_amd.boundary_search_backward(e, tx, ty, 32, r.log_alpha, w)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 0 and float(good.abs().max()) > 0
    assert lib.aligner_debug_set_option(b"mobo_drop_segment", 1) == 0
    try:
        bad = aligner_amd.boundary_search_backward(e, tx, ty, 32, r.log_alpha, w)
        torch.cuda.synchronize()
    finally:
        lib.aligner_debug_set_option(b"mobo_drop_segment", -1)
    assert mobo.read_status(dev) & _lib.ST_INTERNAL
    assert float(bad.abs().max()) == 0
    again = aligner_amd.boundary_search_backward(e, tx, ty, 32, r.log_alpha, w)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 0 and torch.equal(again, good)
