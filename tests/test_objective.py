"""Forward-sum objective, beta-binomial prior and length regulator.

CPU tests pin the oracle's own maths (brute-force path enumeration, autograd, scipy); GPU tests
compare the HIP kernels with the oracle through the C ABI.  Parity is against the build-defined
spec (the reference snapshot does not contain these steps): tolerances are written at each check.
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import forward_sum_oracle as FS


def _rand_logp(rng, B, Tx, Ty, scale=3.0):
    z = rng.standard_normal((B, Tx, Ty)) * scale
    z = z - np.log(np.exp(z).sum(axis=1, keepdims=True))          # log-softmax over the text axis
    return z.astype(np.float32)


# --------------------------------------------------------------------------- oracle (CPU)
def test_oracle_matches_brute_force_enumeration():
    rng = np.random.default_rng(0)
    tx, ty = 3, 6
    lp = _rand_logp(rng, 1, tx, ty)[0].astype(np.float64)
    total, occ = 0.0, np.zeros((tx, ty))
    # a monotonic alignment = the frames at which the path advances: choose tx-1 of the ty-1 steps
    for adv in itertools.combinations(range(1, ty), tx - 1):
        x, w, cells = 0, 0.0, []
        for y in range(ty):
            if y in adv:
                x += 1
            w += lp[x, y]
            cells.append((x, y))
        total += np.exp(w)
        for c in cells:
            occ[c] += np.exp(w)
    logz, post = FS.forward_sum_one(lp, tx, ty)
    assert abs(logz - np.log(total)) < 1e-12
    assert np.abs(post - occ / total).max() < 1e-12


def test_oracle_gradient_is_autograd_of_its_loss():
    rng = np.random.default_rng(1)
    B, Tx, Ty = 2, 5, 9
    lp = _rand_logp(rng, B, Tx, Ty)
    tx, ty = np.array([5, 3]), np.array([9, 7])
    loss, grad = FS.forward_sum(lp, tx, ty)
    t = torch.tensor(lp, dtype=torch.float64, requires_grad=True)
    tot = 0
    for b in range(B):
        NEG = -1e30                                            # (autograd of logaddexp(-inf, -inf) is NaN)
        a = torch.full((int(tx[b]),), NEG, dtype=torch.float64)
        a = torch.cat([t[b, 0:1, 0], a[1:]])
        for y in range(1, int(ty[b])):
            up = torch.cat([torch.tensor([NEG], dtype=torch.float64), a[:-1]])
            a = torch.logaddexp(a, up) + t[b, :int(tx[b]), y]
        tot = tot - a[-1]
        assert abs(float(-a[-1]) - loss[b]) < 1e-10
    tot.backward()
    assert np.abs(t.grad.numpy() - grad).max() < 1e-10
    # every frame of a valid utterance is owned by exactly one token in expectation
    assert np.allclose(-grad[0].sum(axis=0), 1.0, atol=1e-10)


def test_oracle_degenerate_lengths():
    lp = np.zeros((4, 3), np.float32)
    logz, post = FS.forward_sum_one(lp, 4, 3)                    # t_x > t_y: no alignment
    assert logz == -np.inf and not post.any()
    logz, post = FS.forward_sum_one(lp, 1, 3)                    # one token owns everything
    assert logz == 0.0 and np.allclose(post[0], 1.0)


def test_prior_oracle_is_a_distribution_and_follows_the_diagonal():
    pr = FS.beta_binomial_prior(20, 100)
    assert pr.shape == (20, 100)
    # pmf over x = 0..n; the prior keeps x < n, so each frame sums to 1 - pmf(n) (most of it early on)
    assert (pr.sum(axis=0) <= 1.0 + 1e-12).all() and (pr.sum(axis=0)[:50] > 0.9).all()
    centre = (pr * np.arange(20)[:, None]).sum(axis=0) / pr.sum(axis=0)
    assert np.all(np.diff(centre) > 0)                           # the mass moves down the text with time


def test_regulate_oracle():
    h = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    dur = np.array([[2, 0, 1, 3], [1, 1, 1, 1]], np.int32)
    out = FS.regulate(h, dur, 7)
    assert out.shape == (2, 3, 7)
    assert np.array_equal(out[0, 0], [0, 0, 2, 3, 3, 3, 0])
    assert np.array_equal(out[1, 2], [20, 21, 22, 23, 0, 0, 0])


def test_ctc_oracle_is_torch_ctc_loss_and_reduces_to_known_cases():
    """The CTC-form oracle is torch.nn.functional.ctc_loss (float64, CPU).  Sanity of the wiring: one token over T
    frames with a blank so unlikely that it never fires is the plain sum of the frame log-probs, and brute force over
    all labellings of a tiny case."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 1, 5))
    loss, grad = FS.ctc_forward_sum(x, [1], [5], blank_logprob=-200.0)
    assert abs(loss[0]) < 1e-9 and np.abs(grad).max() < 1e-9               # softmax over {blank, token} is ~(0, 1)
    # brute force: K = 2 tokens, T = 4 frames, alphabet {blank, 1, 2}: sum over frame labellings that collapse to (1, 2)
    K, T, blank = 2, 4, -0.7
    x = rng.standard_normal((1, K, T))
    lp = np.concatenate([np.full((1, T), blank), x[0]], 0)
    lp = lp - np.log(np.exp(lp).sum(0, keepdims=True))
    tot = 0.0
    for lab in itertools.product(range(K + 1), repeat=T):
        col = [c for i, c in enumerate(lab) if c != 0 and (i == 0 or lab[i - 1] != c)]
        if col == [1, 2]:
            tot += np.exp(sum(lp[c, t] for t, c in enumerate(lab)))
    loss, _ = FS.ctc_forward_sum(x, [K], [T], blank_logprob=blank)
    assert abs(loss[0] + np.log(tot)) < 1e-10


# --------------------------------------------------------------------------- HIP path (GPU)
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import aligner_amd  # noqa: F401
    from aligner_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


# T_text <= 252 / 504 takes the four- / eight-wave systolic kernels (wave boundaries every 63 rows), wider
# text the one-wave kernels; `one_wave` forces the latter on the small shapes as well
@gpu
@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("B,Tx,Ty,ragged", [(3, 7, 19, True), (4, 64, 200, True), (2, 200, 1000, False),
                                             (2, 300, 700, True), (1, 520, 900, True), (2, 33, 33, False),
                                             (3, 63, 150, True), (3, 127, 333, True), (2, 189, 190, False),
                                             (3, 190, 401, True), (2, 252, 640, True), (2, 253, 500, True), (2, 379, 800, True),
                                             (1, 504, 1100, False), (1, 505, 700, True)])
def test_forward_sum_matches_oracle(dev, request, B, Tx, Ty, ragged, one_wave):
    import aligner_amd
    from aligner_amd import _lib
    if one_wave:
        if Tx > 504:
            pytest.skip("already the one-wave kernel")
        _lib.check(_lib.load().aligner_debug_set_option(b"fwdsum_one_wave", 1))
        request.addfinalizer(lambda: _lib.load().aligner_debug_set_option(b"fwdsum_one_wave", 0))
    rng = np.random.default_rng(B * 1000 + Tx)
    lp = _rand_logp(rng, B, Tx, Ty)
    if ragged:
        ty = rng.integers(max(Tx // 2, 2), Ty + 1, size=B)
        tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
        tx[0], ty[0] = Tx, Ty
    else:
        tx, ty = np.full(B, Tx), np.full(B, Ty)
    want_loss, want_grad = FS.forward_sum(lp, tx, ty)
    loss, grad = aligner_amd.forward_sum(torch.from_numpy(lp).to(dev), torch.from_numpy(tx), torch.from_numpy(ty))
    torch.cuda.synchronize()
    loss, grad = loss.cpu().numpy().astype(np.float64), grad.cpu().numpy().astype(np.float64)
    # The recursion runs in fp32 log space (offsets in double): after T_mel = 1000 frames the
    # log-domain values carry an absolute error of ~1e-4 (measured; an fp32 emulation of the same
    # recursion in numpy shows the same), i.e. a RELATIVE error of that size in every probability.
    #   loss: |d| <= 5e-4 absolute (fp32 output: 2e-7 relative on top)
    #   gradient = -posterior: |d| <= 1e-3 * |oracle| + 2e-5
    assert np.abs(loss - want_loss).max() <= 5e-4 + 2e-7 * np.abs(want_loss).max()
    assert (np.abs(grad - want_grad) <= 1e-3 * np.abs(want_grad) + 2e-5).all()
    for b in range(B):                                          # size-independent property: one token per frame
        assert np.allclose(-grad[b, :, :ty[b]].sum(axis=0), 1.0, atol=1e-3)
        assert not grad[b, tx[b]:].any() and not grad[b, :, ty[b]:].any()


@gpu
def test_forward_sum_degenerate_and_loss_only(dev):
    import aligner_amd
    rng = np.random.default_rng(5)
    lp = _rand_logp(rng, 3, 10, 12)
    tx, ty = np.array([10, 9, 0]), np.array([12, 8, 5])          # ok, t_x > t_y, t_x == 0
    loss, grad = aligner_amd.forward_sum(torch.from_numpy(lp).to(dev), torch.from_numpy(tx), torch.from_numpy(ty))
    loss2, none = aligner_amd.forward_sum(torch.from_numpy(lp).to(dev), torch.from_numpy(tx), torch.from_numpy(ty),
                                          want_grad=False)
    torch.cuda.synchronize()
    want, _ = FS.forward_sum(lp, tx, ty)
    assert none is None and torch.equal(loss, loss2)
    assert abs(float(loss[0]) - want[0]) < 1e-4 and torch.isinf(loss[1]) and torch.isinf(loss[2])
    assert not grad[1].any() and not grad[2].any()


@gpu
def test_forward_sum_is_an_upper_bound_of_the_best_path(dev):
    """log Z >= score of the single best alignment (maximum_path), with equality only for one path."""
    import aligner_amd
    rng = np.random.default_rng(9)
    B, Tx, Ty = 4, 50, 300
    lp = torch.from_numpy(_rand_logp(rng, B, Tx, Ty)).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32)
    ty = torch.full((B,), Ty, dtype=torch.int32)
    loss, _ = aligner_amd.forward_sum(lp, tx, ty, want_grad=False)
    al = aligner_amd.align(lp, tx.to(dev), ty.to(dev))
    best = (al.path * lp).sum(dim=(1, 2))
    assert torch.all(-loss >= best - 1e-3)


@gpu
@pytest.mark.parametrize("scaling", [1.0, 0.5, 2.0])
def test_prior_matches_scipy(dev, scaling):
    import aligner_amd
    tx, ty = np.array([37, 200, 1]), np.array([300, 1000, 5])
    got = aligner_amd.beta_binomial_prior(torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev), 200, 1000,
                                          scaling=scaling).cpu().numpy()
    for b in range(3):
        want = FS.beta_binomial_prior(int(tx[b]), int(ty[b]), scaling)
        blk = got[b, :tx[b], :ty[b]]
        assert np.abs(blk - want).max() < 1e-6 + 1e-5 * want.max()      # fp32 output of a double evaluation
        assert not got[b, tx[b]:].any() and not got[b, :, ty[b]:].any()


@gpu
def test_regulate_matches_oracle_and_the_path(dev):
    import aligner_amd
    rng = np.random.default_rng(3)
    B, C, Tx, Ty = 3, 40, 300, 1111
    h = rng.standard_normal((B, C, Tx)).astype(np.float32)
    dur = rng.integers(0, 8, size=(B, Tx)).astype(np.int32)
    dur[2, 5:] = 0                                               # a short utterance: most frames are padding
    out, tok = aligner_amd.regulate(torch.from_numpy(h).to(dev), torch.from_numpy(dur).to(dev), Ty)
    assert np.array_equal(out.cpu().numpy(), FS.regulate(h, dur, Ty))
    # durations of an actual alignment reproduce its frame -> token map
    lp = torch.from_numpy(_rand_logp(rng, 2, 60, 400)).to(dev)
    tx = torch.tensor([60, 41], dtype=torch.int32, device=dev)
    ty = torch.tensor([400, 333], dtype=torch.int32, device=dev)
    al = aligner_amd.align(lp, tx, ty, want_tok=True)
    _, tok2 = aligner_amd.regulate(torch.zeros((2, 1, 60), device=dev), al.durations, 400)
    assert torch.equal(tok2, al.tok)


# T_text + 1 <= 252 / 504 rows (the blank after the last token takes one) run on the four- / eight-wave systolic kernels,
# wider text on the one-wave kernels; `one_wave` forces the latter on the small shapes as well
@gpu
@pytest.mark.parametrize("one_wave", [False, True])
@pytest.mark.parametrize("B,Tx,Ty,ragged", [(3, 7, 19, True), (2, 1, 9, False), (4, 64, 200, True), (2, 200, 1000, False),
                                             (2, 255, 600, True), (2, 256, 500, True), (1, 400, 900, True), (2, 33, 33, False),
                                             (3, 62, 150, True), (3, 63, 150, True), (2, 125, 333, True), (2, 126, 300, False),
                                             (2, 251, 640, True), (2, 252, 500, True), (1, 503, 1100, False), (1, 504, 700, True),
                                             (1, 600, 700, True)])
@pytest.mark.parametrize("blank", [-1.0, -6.0])
def test_forward_sum_ctc_form_matches_torch_ctc_loss(dev, request, B, Tx, Ty, ragged, blank, one_wave):
    """The published (CTC / blank) form of the forward-sum loss against torch.nn.functional.ctc_loss in float64 on the
    CPU -- an external implementation -- loss and gradient, through the C ABI, with the tolerances of the plain form."""
    import aligner_amd
    from aligner_amd import _lib
    if one_wave:
        if Tx + 1 > 504:
            pytest.skip("already the one-wave kernel")
        _lib.check(_lib.load().aligner_debug_set_option(b"fwdsum_one_wave", 1))
        request.addfinalizer(lambda: _lib.load().aligner_debug_set_option(b"fwdsum_one_wave", 0))
    rng = np.random.default_rng(B * 777 + Tx)
    x = _rand_logp(rng, B, Tx, Ty)                                    # log-softmax'd attention, as the OTA model feeds it
    if ragged:
        ty = rng.integers(max(Tx // 2, 2), Ty + 1, size=B)
        tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
        tx[0], ty[0] = Tx, Ty
    else:
        tx, ty = np.full(B, Tx), np.full(B, Ty)
    want_loss, want_grad = FS.ctc_forward_sum(x, tx, ty, blank)
    loss, grad = aligner_amd.forward_sum(torch.from_numpy(x).to(dev), torch.from_numpy(tx), torch.from_numpy(ty),
                                         blank_logprob=blank)
    torch.cuda.synchronize()
    loss, grad = loss.cpu().numpy().astype(np.float64), grad.cpu().numpy().astype(np.float64)
    # loss = -(log Z of the raw scores - sum of the frames' normalisers): two sums of ~T_mel fp32 log2 / exp2 results of
    # magnitude ~|loss| each, so the absolute error scales with |loss|: 5e-4 + 1e-6 * |loss| (measured 1.1e-3 at 1 847)
    assert (np.abs(loss - want_loss) <= 5e-4 + 1e-6 * np.abs(want_loss)).all()
    # gradient = softmax - occupancy: the fp32 log-domain recursion carries a RELATIVE error in the occupancy (a number
    # up to 1) that grows with the depth of the lattice -- two states per token, three-way sums: measured (tools/ctc_err.py)
    # 7e-4 at [200,1000], 1.5e-3 at [255,600], 2.9e-3 at [400,900], 4.3e-3 at [600,700] -- so the difference is held to
    # 5e-3 * occupancy + 2e-5 (the occupancy from the oracle)
    x64 = x.astype(np.float64)
    for b in range(B):
        K, T = int(tx[b]), int(ty[b])
        z = np.concatenate([np.full((1, T), blank), x64[b, :K, :T]], 0)
        p = np.exp(z - np.log(np.exp(z).sum(0, keepdims=True)))[1:]
        occ = p - want_grad[b, :K, :T]
        assert occ.min() > -1e-9
        assert (np.abs(grad[b, :K, :T] - want_grad[b, :K, :T]) <= 5e-3 * occ + 2e-5).all(), b
    for b in range(B):
        assert not grad[b, tx[b]:].any() and not grad[b, :, ty[b]:].any()
        # per frame: softmax mass of the text rows minus the tokens' occupancy = occupancy(blank) - softmax(blank)
        assert np.all(np.abs(grad[b, :tx[b], :ty[b]].sum(axis=0)) < 1.0 + 1e-3)


@gpu
def test_forward_sum_ctc_form_degenerate_and_loss_only(dev):
    import aligner_amd
    rng = np.random.default_rng(6)
    x = _rand_logp(rng, 3, 10, 12)
    tx, ty = np.array([10, 9, 0]), np.array([12, 8, 5])          # ok, fewer frames than tokens, no tokens
    loss, grad = aligner_amd.forward_sum(torch.from_numpy(x).to(dev), torch.from_numpy(tx), torch.from_numpy(ty),
                                         blank_logprob=-1.0)
    loss2, none = aligner_amd.forward_sum(torch.from_numpy(x).to(dev), torch.from_numpy(tx), torch.from_numpy(ty),
                                          want_grad=False, blank_logprob=-1.0)
    torch.cuda.synchronize()
    want, wgrad = FS.ctc_forward_sum(x, tx, ty, -1.0)
    assert none is None and torch.equal(loss, loss2)
    assert abs(float(loss[0]) - want[0]) < 1e-4 and torch.isinf(loss[1]) and torch.isinf(loss[2])
    assert not grad[1].any() and not grad[2].any()
    assert np.abs(grad[0].cpu().numpy() - wgrad[0]).max() < 1e-4


# With the gradient asked for, a batch that leaves the CUs half idle runs both sweeps in ONE launch (alpha and beta do not
# read each other) and a combining pass; larger batches, or the debug option, run forward then backward.  Same numbers up
# to fp32 rounding of the exponent's sum; degenerate utterances and the padding are zeros either way.
@gpu
@pytest.mark.parametrize("blank", [None, -1.0])
@pytest.mark.parametrize("B,Tx,Ty", [(5, 40, 130), (3, 200, 1000), (2, 251, 517), (2, 300, 702), (128, 12, 48), (130, 20, 64)])
def test_forward_sum_side_by_side_sweeps_equal_the_serial_form(dev, request, B, Tx, Ty, blank):
    import aligner_amd
    from aligner_amd import _lib
    rng = np.random.default_rng(B + 31 * Tx)
    lp = torch.from_numpy(_rand_logp(rng, B, Tx, Ty)).to(dev)
    ty = rng.integers(max(Tx // 2, 2), Ty + 1, size=B)
    tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
    tx[0], ty[0] = Tx, Ty
    if B > 2:
        tx[1], ty[1] = min(Tx, 9), 5                              # fewer frames than tokens: loss +inf, gradient 0
    tx, ty = torch.from_numpy(tx), torch.from_numpy(ty)
    lib = _lib.load()
    request.addfinalizer(lambda: lib.aligner_debug_set_option(b"fwdsum_serial", 0))
    _lib.check(lib.aligner_debug_set_option(b"fwdsum_serial", 1))
    loss_s, grad_s = aligner_amd.forward_sum(lp, tx, ty, blank_logprob=blank)
    grad_s = grad_s.clone()
    _lib.check(lib.aligner_debug_set_option(b"fwdsum_serial", 0))
    loss_p, grad_p = aligner_amd.forward_sum(lp, tx, ty, blank_logprob=blank)
    torch.cuda.synchronize()
    assert torch.equal(loss_s, loss_p)                            # the forward sweep is the same code
    # (both forms add alpha, beta and the offsets in fp32, in a different order: at |alpha| ~ 300 below its wave's column
    # maximum one rounding is 3e-5 of the posterior -- a fifth of the tolerance against the oracle)
    # (the CTC form's gradient is softmax - occupancy: the same relative rounding of an occupancy of up to 1, absolute)
    tol_abs = 2e-6 if blank is None else 2e-4
    assert bool(((grad_s - grad_p).abs() <= 2e-4 * grad_s.abs() + tol_abs).all()), float((grad_s - grad_p).abs().max())
    for b in range(B):
        assert not grad_p[b, int(tx[b]):].any() and not grad_p[b, :, int(ty[b]):].any()
    if B > 2:
        assert not grad_p[1].any() and torch.isinf(loss_p[1])


# Batches past half the CU count (and `fwdsum_serial`) run forward, then the gradient-making backward kernel, whose stager
# (log-probs + alpha + the forward offsets per tile) issues its loads by hand too (fs_grad_stager_by_hand): the same tiles
# reach LDS as with the compiler-scheduled stager (`fwdsum_no_grad_stager`), so the two must agree BIT FOR BIT; and against
# the oracle.  Shapes: a partial last tile (T_mel % 16 != 0), ragged lengths, text ending inside a wave, both widths of
# the systolic kernels (T_text <= 252: four waves, 16-frame tiles; <= 504: eight waves, 8-frame tiles), [160,200,1000].
@gpu
@pytest.mark.parametrize("B,Tx,Ty", [(3, 200, 1000), (5, 40, 132), (2, 251, 520), (2, 300, 704), (4, 504, 808), (130, 20, 64),
                                     (160, 200, 1000)])
def test_gradient_stager_by_hand(dev, request, B, Tx, Ty):
    import aligner_amd
    from aligner_amd import _lib
    rng = np.random.default_rng(7 * B + Tx)
    lp_h = _rand_logp(rng, B, Tx, Ty)
    lp = torch.from_numpy(lp_h).to(dev)
    ty = rng.integers(max(Tx // 2, 2), Ty + 1, size=B)
    tx = np.minimum(rng.integers(1, Tx + 1, size=B), ty)
    tx[0], ty[0] = Tx, Ty
    if B > 2:
        tx[1], ty[1] = min(Tx, 9), 5                              # fewer frames than tokens: loss +inf, gradient 0
    lib = _lib.load()
    request.addfinalizer(lambda: (lib.aligner_debug_set_option(b"fwdsum_serial", 0),
                                  lib.aligner_debug_set_option(b"fwdsum_no_grad_stager", 0)))
    _lib.check(lib.aligner_debug_set_option(b"fwdsum_serial", 1))
    loss_h, grad_h = aligner_amd.forward_sum(lp, torch.from_numpy(tx), torch.from_numpy(ty))
    grad_h = grad_h.clone()
    _lib.check(lib.aligner_debug_set_option(b"fwdsum_no_grad_stager", 1))
    loss_c, grad_c = aligner_amd.forward_sum(lp, torch.from_numpy(tx), torch.from_numpy(ty))
    torch.cuda.synchronize()
    assert torch.equal(loss_h, loss_c) and torch.equal(grad_h.view(torch.int32), grad_c.view(torch.int32))
    nb = min(B, 4)                                                # the float64 oracle on the first utterances
    want_loss, want_grad = FS.forward_sum(lp_h[:nb], tx[:nb], ty[:nb])
    g = grad_h[:nb].cpu().numpy().astype(np.float64)
    fin = np.isfinite(want_loss)
    assert np.abs(loss_h[:nb].cpu().numpy().astype(np.float64)[fin] - want_loss[fin]).max() <= 5e-4 + 2e-7 * np.abs(want_loss[fin]).max()
    assert (np.abs(g - want_grad) <= 1e-3 * np.abs(want_grad) + 2e-5).all()
    for b in range(B):
        assert not grad_h[b, int(tx[b]):].any() and not grad_h[b, :, int(ty[b]):].any()


# The systolic kernels' stagers issue 16-byte global accesses by hand when T_mel % 4 == 0 and the tensors are 16-byte
# aligned; anything else takes the compiler-scheduled stager.  A log-prob tensor that starts 4 bytes into its buffer must
# give the same numbers (and no misaligned access).
@gpu
@pytest.mark.parametrize("blank", [None, -1.0])
def test_forward_sum_on_a_tensor_that_is_not_16_byte_aligned(dev, blank):
    import aligner_amd
    rng = np.random.default_rng(77)
    B, Tx, Ty = 3, 100, 240
    lp = torch.from_numpy(_rand_logp(rng, B, Tx, Ty)).to(dev)
    buf = torch.empty(B * Tx * Ty + 1, dtype=torch.float32, device=dev)
    off = buf[1:].view(B, Tx, Ty)
    off.copy_(lp)
    assert off.data_ptr() % 16 == 4 and off.is_contiguous()
    tx = torch.tensor([Tx, 37, 90], dtype=torch.int32)
    ty = torch.tensor([Ty, 200, 91], dtype=torch.int32)
    loss_a, grad_a = aligner_amd.forward_sum(lp, tx, ty, blank_logprob=blank)
    loss_o, grad_o = aligner_amd.forward_sum(off, tx, ty, blank_logprob=blank)
    torch.cuda.synchronize()
    assert torch.allclose(loss_a, loss_o, rtol=1e-6, atol=1e-5)
    assert bool(((grad_a - grad_o).abs() <= 2e-4 * grad_a.abs() + (2e-6 if blank is None else 2e-4)).all())


@gpu
@pytest.mark.parametrize("blank", [None, -1.0])
def test_forward_sum_loss_is_differentiable_like_its_float64_restatement(dev, blank):
    """forward_sum_loss (the OTA ForwardSumLoss on the kernels, attached to autograd): the gradient that reaches a leaf
    BEHIND a log_softmax equals torch autograd through the float64 oracle restatement (the CTC form: torch's own ctc_loss)."""
    import aligner_amd
    rng = np.random.default_rng(3)
    B, Tx, Ty = 3, 24, 60
    z0 = rng.standard_normal((B, Tx, Ty))
    tx = torch.tensor([Tx, 17, 9], dtype=torch.int32)
    ty = torch.tensor([Ty, 44, 30], dtype=torch.int32)
    z = torch.tensor(z0, dtype=torch.float32, device=dev, requires_grad=True)
    loss = aligner_amd.forward_sum_loss(torch.log_softmax(z, dim=1), tx, ty, blank_logprob=blank, reduction="sum")
    loss.backward()
    zr = torch.tensor(z0, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(zr, dim=1)
    lpn = lp.detach().numpy()
    if blank is None:
        want_loss, want_grad = FS.forward_sum(lpn.astype(np.float32), tx.numpy(), ty.numpy())
    else:
        want_loss, want_grad = FS.ctc_forward_sum(lpn.astype(np.float32), tx.numpy(), ty.numpy(), blank)
    lp.backward(gradient=torch.from_numpy(np.asarray(want_grad, dtype=np.float64)))     # the chain rule through log_softmax
    total = float(np.sum(want_loss))
    assert abs(float(loss.detach()) - total) <= 2e-3 + 1e-6 * abs(total)
    assert torch.allclose(z.grad.cpu().double(), zr.grad, rtol=2e-3, atol=5e-5)
    # no gradient asked for: the loss alone (one sweep)
    l2 = aligner_amd.forward_sum_loss(torch.log_softmax(z.detach(), dim=1), tx, ty, blank_logprob=blank, reduction="none")
    assert not l2.requires_grad and l2.shape == (B,)


@gpu
def test_forward_sum_loss_options_and_input_dtype(dev):
    """forward_sum_loss's CTCLoss-style switches and a 16-bit input: an infeasible utterance (t_x > t_y) makes the plain batch
    loss +inf and, with zero_infinity, contributes 0 and no gradient; length_normalize divides each loss by its t_x; a bf16
    logp gets its gradient back as bf16 (the kernels compute in fp32)."""
    import aligner_amd
    rng = np.random.default_rng(8)
    B, Tx, Ty = 3, 20, 48
    z0 = torch.tensor(rng.standard_normal((B, Tx, Ty)), dtype=torch.float32, device=dev)
    tx = torch.tensor([Tx, 12, 20], dtype=torch.int32)
    ty = torch.tensor([Ty, 30, 10], dtype=torch.int32)           # the last one: 20 tokens on 10 frames
    z = z0.clone().requires_grad_(True)
    each = aligner_amd.forward_sum_loss(torch.log_softmax(z, dim=1), tx, ty, reduction="none")
    assert bool(torch.isinf(each[2])) and bool(torch.isfinite(each[:2]).all())
    assert bool(torch.isinf(aligner_amd.forward_sum_loss(torch.log_softmax(z, dim=1), tx, ty)))
    safe = aligner_amd.forward_sum_loss(torch.log_softmax(z, dim=1), tx, ty, reduction="sum", zero_infinity=True)
    assert abs(float(safe.detach()) - float(each[:2].detach().sum())) < 1e-3
    safe.backward()
    assert bool(torch.isfinite(z.grad).all()) and float(z.grad[2].abs().max()) == 0.0 and float(z.grad[0].abs().max()) > 0
    norm = aligner_amd.forward_sum_loss(torch.log_softmax(z.detach(), dim=1), tx, ty, reduction="none", zero_infinity=True,
                                        length_normalize=True)
    assert torch.allclose(norm[:2].cpu(), (each[:2].detach().cpu() / tx[:2].float()), rtol=1e-6)
    # 16-bit log-probs
    zb = z0[:2].to(torch.bfloat16).requires_grad_(True)
    lb = aligner_amd.forward_sum_loss(torch.log_softmax(zb.float(), dim=1).to(torch.bfloat16), tx[:2], ty[:2], reduction="sum")
    lb.backward()
    assert zb.grad.dtype == torch.bfloat16 and bool(torch.isfinite(zb.grad).all()) and float(zb.grad.abs().max()) > 0
