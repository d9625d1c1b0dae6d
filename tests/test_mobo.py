"""MoBoAligner monotonic boundary search (BASELINE config 5).

Parity UNPINNED (the reference snapshot holds only the branch name and the paper link, README.md:9-13,49):
the float64 oracle oracle/mobo_oracle.py is this build's restatement of the paper's formulation, pinned here to
brute-force enumeration of every boundary sequence on tiny shapes; the HIP path is checked against the oracle."""
import numpy as np
import pytest
import torch

from oracle import mobo_oracle as M

gpu = pytest.mark.gpu


def test_oracle_matches_brute_force_enumeration():
    rng = np.random.default_rng(3)
    n = 0
    for (I, J, D) in [(1, 1, 1), (1, 3, 3), (2, 3, 2), (3, 5, 3), (3, 7, 3), (4, 6, 2), (4, 9, 4), (5, 8, 3), (3, 9, 3),
                      (2, 8, 7), (6, 6, 3)]:
        for rep in range(3):
            e = rng.standard_normal((I, J)) * (1.0 + 2.0 * rep)
            got, want = M.boundary_search(e, D), M.brute_force(e, D)
            fin = np.isfinite(want["log_alpha"])
            assert np.array_equal(np.isfinite(got["log_alpha"]), fin)
            assert np.allclose(got["log_alpha"][fin], want["log_alpha"][fin], atol=1e-10)
            assert np.allclose(got["gamma"], want["gamma"], atol=1e-10)
            assert np.array_equal(got["boundaries"], want["boundaries"])
            assert abs(got["map_score"] - want["map_score"]) < 1e-10
            assert abs(M.sequence_log_prob(e, D, got["boundaries"]) - got["map_score"]) < 1e-10
            n += 1
    assert n == 33


def test_oracle_properties_at_moderate_size():
    rng = np.random.default_rng(4)
    I, J, D = 40, 300, 16
    r = M.boundary_search(rng.standard_normal((I, J)) * 2, D)
    assert np.allclose(np.exp(r["log_alpha"]).sum(1), 1.0, atol=1e-9)        # every boundary is somewhere
    assert np.allclose(r["gamma"].sum(0), 1.0, atol=1e-9)                    # every frame belongs to one token
    assert r["gamma"].min() > -1e-12
    d = r["durations"]
    assert d.sum() == J and d.min() >= 1 and d.max() <= D and r["boundaries"][-1] == J
    with pytest.raises(ValueError):
        M.boundary_search(np.zeros((3, 10)), 3)                              # 3 tokens of <= 3 frames cannot cover 10


def test_vectorised_oracle_equals_the_plain_loops():
    """boundary_search_fast (numpy sliding windows, what the full-size GPU comparison uses) against the loops."""
    rng = np.random.default_rng(5)
    for (I, J, D) in [(1, 1, 1), (2, 3, 2), (5, 8, 3), (12, 40, 8), (30, 200, 7), (9, 90, 10), (40, 300, 16), (64, 257, 32),
                      (6, 70, 64)]:
        e = rng.standard_normal((I, J)) * 3
        e[rng.random((I, J)) < 0.02] = -np.inf                                 # masked frames
        a, f = M.boundary_search(e, D), M.boundary_search_fast(e, D)
        fin = np.isfinite(a["log_alpha"])
        assert np.array_equal(fin, np.isfinite(f["log_alpha"]))
        assert np.allclose(a["log_alpha"][fin], f["log_alpha"][fin], atol=1e-11)
        assert np.allclose(a["gamma"], f["gamma"], atol=1e-12)
        assert np.array_equal(a["boundaries"], f["boundaries"])
        assert a["map_score"] == f["map_score"] or abs(a["map_score"] - f["map_score"]) < 1e-11


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _check_against_oracle(dev, e, tx, ty, D, dt=torch.float32):
    import aligner_amd
    ed = torch.from_numpy(e).to(dt)
    r = aligner_amd.boundary_search(ed.to(dev), torch.from_numpy(tx), torch.from_numpy(ty), D, want_log_alpha=True,
                                    want_gamma=True)
    torch.cuda.synchronize()
    la, ga = r.log_alpha.cpu().numpy().astype(np.float64), r.gamma.cpu().numpy().astype(np.float64)
    bnd, dur, sc = r.boundaries.cpu().numpy(), r.durations.cpu().numpy(), r.map_score.cpu().numpy()
    e64 = ed.float().numpy().astype(np.float64)
    for b in range(e.shape[0]):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_fast(e64[b, :I, :J], D)
        fin = np.isfinite(want["log_alpha"])
        assert np.array_equal(np.isfinite(la[b, :I, :J]), fin), b
        assert np.all(np.isneginf(la[b, I:])) and np.all(np.isneginf(la[b, :I, J:]))
        # fp32 log-domain recursion over I rows: absolute error grows with the depth of the chain
        assert np.abs(la[b, :I, :J][fin] - want["log_alpha"][fin]).max() < 2e-3 + 2e-5 * I, b
        assert np.abs(ga[b, :I, :J] - want["gamma"]).max() < 2e-4 + 2e-6 * I, b
        assert np.all(ga[b, I:] == 0) and np.all(ga[b, :, J:] == 0)
        # the MAP sequence: a valid segmentation whose log-probability (evaluated in float64) is the optimum
        bb = bnd[b, :I]
        assert np.array_equal(np.diff(np.concatenate([[0], bb])), dur[b, :I]) and bb[-1] == J
        assert dur[b, :I].min() >= 1 and dur[b, :I].max() <= D and np.all(bnd[b, I:] == J)
        lp = M.sequence_log_prob(e64[b, :I, :J], D, bb)
        assert lp >= want["map_score"] - 1e-3 - 1e-5 * I, (b, lp, want["map_score"])
        assert abs(sc[b] - lp) < 2e-3 + 2e-5 * I


@gpu
@pytest.mark.parametrize("B,Tx,Ty,D", [(3, 1, 1, 1), (2, 5, 9, 3), (4, 12, 40, 8), (3, 40, 300, 16), (2, 64, 257, 32),
                                       (2, 30, 1100, 64), (1, 100, 600, 7), (2, 9, 90, 10),
                                       (1, 120, 1000, 16),      # one utterance over 16 position segments
                                       (5, 33, 700, 40),        # ragged utterances: fewer segments than the launch has
                                       (2, 8, 1500, 800),       # a window of half the utterance: one segment, 1501 positions
                                       (1, 6, 2600, 1300),      # ... several positions per thread (state in LDS)
                                       (300, 4, 20, 6)])        # more utterances than CUs
def test_boundary_search_matches_oracle(dev, B, Tx, Ty, D):
    rng = np.random.default_rng(B * 100 + Tx)
    e = (rng.standard_normal((B, Tx, Ty)) * 2).astype(np.float32)
    tx = np.array([Tx] + [int(rng.integers(max(1, -(-Ty // (2 * D))), Tx + 1)) for _ in range(B - 1)], np.int32)
    ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, B)], np.int32)
    if Ty > Tx * D:
        ty[0] = Tx * D
    _check_against_oracle(dev, e, tx, ty, D)


@gpu
@pytest.mark.parametrize("B,Tx,Ty,D", [(3, 1, 1, 1), (2, 5, 9, 3), (4, 12, 40, 8), (3, 40, 300, 16), (2, 64, 257, 32),
                                       (2, 30, 1100, 64), (1, 100, 600, 7), (1, 120, 1000, 16), (5, 33, 700, 40),
                                       (2, 8, 1500, 800), (8, 500, 4000, 32), (300, 4, 20, 6)])
def test_map_only_chain_equals_the_full_chain(dev, B, Tx, Ty, D):
    """Without log_alpha the search runs the max-product chain alone (mobo_chain_map_kernel); the same arithmetic as the
    max-product half of the full kernel, so boundaries, durations and the score are bit for bit the full kernel's
    (`mobo_full_chain` forces that one), ragged batches included -- and the score is the sequence's log-probability."""
    import aligner_amd
    from aligner_amd import _lib, mobo
    lib = _lib.load()
    rng = np.random.default_rng(B * 10 + Tx + D)
    e = (rng.standard_normal((B, Tx, Ty)) * 2).astype(np.float32)
    tx = np.array([Tx] + [int(rng.integers(max(1, -(-Ty // (2 * D))), Tx + 1)) for _ in range(B - 1)], np.int32)
    ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, B)], np.int32)
    if Ty > Tx * D:
        ty[0] = Tx * D
    ed = torch.from_numpy(e).to(dev)
    got = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D)
    assert lib.aligner_debug_set_option(b"mobo_full_chain", 1) == 0
    try:
        ref = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D)
        torch.cuda.synchronize()
    finally:
        lib.aligner_debug_set_option(b"mobo_full_chain", 0)
    assert mobo.read_status(dev) == 0
    assert torch.equal(got.boundaries, ref.boundaries) and torch.equal(got.durations, ref.durations)
    assert torch.equal(got.map_score, ref.map_score)
    dur = got.durations.cpu().numpy()
    for b in range(min(B, 4)):
        I, J = int(tx[b]), int(ty[b])
        assert dur[b, :I].min() >= 1 and dur[b, :I].max() <= D and dur[b, :I].sum() == J
        if I * J <= 200000:
            lp = M.sequence_log_prob(e[b, :I, :J].astype(np.float64), D, got.boundaries[b, :I].cpu().numpy())
            assert abs(float(got.map_score[b]) - lp) < 2e-3 + 2e-5 * I


@gpu
@pytest.mark.parametrize("B,Tx,Ty,D", [(3, 1, 1, 1), (2, 5, 9, 3), (4, 12, 40, 8), (3, 40, 300, 16), (2, 64, 257, 32),
                                       (2, 30, 1100, 64), (1, 100, 600, 7), (1, 120, 1000, 16), (5, 33, 700, 40),
                                       (2, 8, 1500, 800), (1, 6, 2600, 1300), (300, 4, 20, 6)])
@pytest.mark.parametrize("scale", [2.0, 30.0])
def test_sum_only_chain_equals_the_full_chain(dev, B, Tx, Ty, D, scale):
    """want_map=False runs the sum-product chain alone (mobo_chain_sum_kernel): log_alpha and gamma bit for bit the full
    kernel's, also where rows take the exact sums (energies of +-60 nats), and no MAP output is touched."""
    import aligner_amd
    from aligner_amd import mobo
    rng = np.random.default_rng(B * 10 + Tx + D)
    e = (rng.standard_normal((B, Tx, Ty)) * scale).astype(np.float32)
    tx = np.array([Tx] + [int(rng.integers(max(1, -(-Ty // (2 * D))), Tx + 1)) for _ in range(B - 1)], np.int32)
    ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, B)], np.int32)
    if Ty > Tx * D:
        ty[0] = Tx * D
    ed = torch.from_numpy(e).to(dev)
    full = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D, want_gamma=True)
    only = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D, want_gamma=True, want_map=False)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 0
    assert only.boundaries is None and only.durations is None and only.map_score is None
    assert torch.equal(only.log_alpha, full.log_alpha) and torch.equal(only.gamma, full.gamma)
    with pytest.raises(ValueError):
        aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D, want_map=False)


@gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_boundary_search_sixteen_bit_energies(dev, dt):
    rng = np.random.default_rng(8)
    e = (rng.standard_normal((2, 20, 150)) * 2).astype(np.float32)
    _check_against_oracle(dev, e, np.array([20, 13], np.int32), np.array([150, 90], np.int32), 16, dt)


@gpu
def test_boundary_search_infeasible_and_properties_at_config5_size(dev):
    """BASELINE config 5 at full size [8, T_text=500, T_mel=4000], bf16 scores in, int32 boundaries out: two whole
    utterances (the full-length one and a ragged one) against the oracle, properties that need no oracle for all
    eight (every row of alpha and every column of gamma sums to 1, durations within the window), and the
    infeasible case."""
    import aligner_amd
    from aligner_amd import mobo
    g = torch.Generator().manual_seed(12)
    B, Tx, Ty, D = 8, 500, 4000, 32
    e = (torch.randn(B, Tx, Ty, generator=g) * 2).bfloat16()
    tx = torch.tensor([500, 480, 333, 250, 500, 125, 412, 499], dtype=torch.int32)
    ty = torch.tensor([4000, 3999, 2800, 2100, 3504, 4000, 3333, 4000], dtype=torch.int32)
    r = aligner_amd.boundary_search(e.to(dev), tx, ty, D, want_gamma=True)
    torch.cuda.synchronize()
    la, ga, dur, bnd = r.log_alpha.cpu(), r.gamma.cpu(), r.durations.cpu().numpy(), r.boundaries.cpu().numpy()
    for b in range(B):
        I, J = int(tx[b]), int(ty[b])
        assert torch.allclose(torch.exp(la[b, :I, :J].double()).sum(1), torch.ones(I, dtype=torch.float64), atol=2e-3)
        assert torch.allclose(ga[b, :I, :J].sum(0), torch.ones(J), atol=2e-3)
        assert float(ga[b].min()) > -1e-3
        assert dur[b, :I].sum() == J and dur[b, :I].min() >= 1 and dur[b, :I].max() <= D and bnd[b, I - 1] == J
    assert mobo.read_status(dev) == 0
    e64 = e.float().numpy().astype(np.float64)
    for b in (0, 2):
        I, J = int(tx[b]), int(ty[b])
        want = M.boundary_search_fast(e64[b, :I, :J], D)
        fin = np.isfinite(want["log_alpha"])
        got = la[b, :I, :J].numpy().astype(np.float64)
        assert np.array_equal(np.isfinite(got), fin)
        assert np.abs(got[fin] - want["log_alpha"][fin]).max() < 2e-3 + 2e-5 * I
        assert np.abs(ga[b, :I, :J].numpy() - want["gamma"]).max() < 2e-4 + 2e-6 * I
        lp = M.sequence_log_prob(e64[b, :I, :J], D, bnd[b, :I])
        assert lp >= want["map_score"] - 1e-3 - 1e-5 * I
        assert abs(float(r.map_score[b]) - lp) < 2e-3 + 2e-5 * I
    # an utterance no segmentation can cover
    r = aligner_amd.boundary_search(e[:2, :10, :400].to(dev), torch.tensor([10, 10]), torch.tensor([400, 300]), 32)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) & 1
    assert r.durations[0].sum().item() == 0 and r.durations[1].sum().item() == 300


@gpu
def test_boundary_search_deep_tail_states_keep_their_precision(dev):
    """Energies of +-60 nats: most windows lie far below their row's bulk (states hundreds of nats down).  Every
    window is summed against its own maximum, so log_alpha holds there too."""
    rng = np.random.default_rng(21)
    e = (rng.standard_normal((2, 40, 500)) * 30).astype(np.float32)
    _check_against_oracle(dev, e, np.array([40, 31], np.int32), np.array([500, 420], np.int32), 24)


@gpu
def test_boundary_search_segment_that_never_delivers_fails_loudly(dev):
    """The position segments of an utterance wait for one another through the workspace.  If one never delivers
    (here: switched off by the test hook) its successor gives up after a bounded wait, the call ends, the status
    word says ALIGNER_ST_INTERNAL and the utterance's boundaries / durations are all zero -- never a wrong answer."""
    import aligner_amd
    from aligner_amd import _lib, mobo
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    e = torch.randn(2, 50, 900, generator=g).to(dev)
    tx, ty = torch.tensor([50, 40]), torch.tensor([900, 700])
    good = aligner_amd.boundary_search(e, tx, ty, 32)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 0 and int(good.durations[0].sum()) == 900
    assert lib.aligner_debug_set_option(b"mobo_drop_segment", 1) == 0
    try:
        r = aligner_amd.boundary_search(e, tx, ty, 32)
        torch.cuda.synchronize()
    finally:
        lib.aligner_debug_set_option(b"mobo_drop_segment", -1)
    assert mobo.read_status(dev) & _lib.ST_INTERNAL
    assert int(r.durations.abs().sum()) == 0 and int(r.boundaries.abs().sum()) == 0
    assert bool(torch.isneginf(r.map_score).all())
    again = aligner_amd.boundary_search(e, tx, ty, 32)          # and the next call is fine
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 0 and torch.equal(again.boundaries, good.boundaries)


@gpu
def test_boundary_search_masked_frames_without_any_segmentation(dev):
    """Masked (-inf) energies can leave no boundary sequence a positive probability (here: every token's last admissible
    frame is masked in a tight utterance).  That is an input without an answer, not an internal error: all-zero
    boundaries / durations, map_score -inf, ALIGNER_ST_BAD_LENGTHS -- and the batch's other utterances are untouched.
    (Found by tools/soak_mobo.py: the first build reported ALIGNER_ST_INTERNAL.)"""
    import aligner_amd
    from aligner_amd import mobo
    rng = np.random.default_rng(31)
    B, Tx, Ty, D = 3, 4, 8, 2
    e = rng.standard_normal((B, Tx, Ty)).astype(np.float32)
    e[1, 3, 7] = -np.inf                                  # 4 tokens x 2 frames = 8: token 3 must end at frame 7
    tx, ty = np.array([4, 4, 4], np.int32), np.array([8, 8, 7], np.int32)
    assert not np.isfinite(M.boundary_search_fast(e[1].astype(np.float64), D)["map_score"])
    r = aligner_amd.boundary_search(torch.from_numpy(e).to(dev), torch.from_numpy(tx), torch.from_numpy(ty), D, want_gamma=True)
    torch.cuda.synchronize()
    assert mobo.read_status(dev) == 1
    assert not r.durations[1].any() and not r.boundaries[1].any() and torch.isneginf(r.map_score[1])
    for b in (0, 2):
        want = M.boundary_search_fast(e[b, :tx[b], :ty[b]].astype(np.float64), D)
        assert np.array_equal(r.boundaries[b].cpu().numpy()[:tx[b]], want["boundaries"])
        assert abs(float(r.map_score[b]) - want["map_score"]) < 1e-3


@gpu
def test_sharded_boundary_search_product_path_single_rank(dev):
    """The multi-GPU entry point with the boundary search as the per-rank compute (sharded.boundary_durations) on a 1-rank
    RCCL group: LPT order, the gather and the un-permutation around the HIP kernels; durations equal a direct call."""
    import os
    import torch.distributed as dist
    import aligner_amd
    from aligner_amd import sharded
    rng = np.random.default_rng(77)
    n, Tx, Ty, D = 11, 40, 300, 16
    e = (rng.standard_normal((n, Tx, Ty)) * 2).astype(np.float32)
    tx = np.array([Tx] + [int(rng.integers(20, Tx + 1)) for _ in range(n - 1)], np.int32)
    ty = np.array([Ty] + [int(rng.integers(tx[b], min(Ty, tx[b] * D) + 1)) for b in range(1, n)], np.int32)
    ed = torch.from_numpy(e).to(dev)
    want = aligner_amd.boundary_search(ed, torch.from_numpy(tx), torch.from_numpy(ty), D).durations
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        got = sharded.sharded_align(lambda idx: ed[torch.as_tensor(idx, device=dev)], tx, ty, Tx,
                                    align_fn=sharded.boundary_durations(D), device=dev)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert torch.equal(got, want) and bool((got.sum(1).cpu() == torch.from_numpy(ty)).all())
