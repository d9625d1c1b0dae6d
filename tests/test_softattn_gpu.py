"""GPU tests of the soft-attention front end against oracle/softattn_oracle.py
(fp32 torch on CPU).  Tolerance 1e-4 absolute on log-probabilities, the figure
BASELINE.json's north_star states; parity is UNPINNED (source absent from the
reference snapshot, see the oracle's header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _cmp(got, want, tol=TOL):
    got = got.cpu()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin)
    assert torch.equal(got[~fin], want[~fin])
    return (got[fin] - want[fin]).abs().max().item()


@pytest.mark.parametrize("B,C,Tx,Ty,sim", [(2, 80, 50, 130, "l2"), (3, 80, 200, 333, "l2"), (2, 16, 7, 40, "l2"),
                                           (2, 80, 224, 129, "dot"), (2, 96, 100, 260, "l2"), (1, 80, 300, 257, "l2"),
                                           (2, 80, 500, 700, "l2"), (1, 200, 130, 64, "l2"),
                                           # C == 16*KS for KS = 8 (hand-issued loads), KS = 16 in one row group
                                           # (two staging passes), B a multiple of 8 (XCD-aware workgroup map)
                                           (8, 128, 64, 300, "l2"), (1, 256, 100, 96, "dot"), (16, 80, 40, 600, "l2")])
def test_logp_matches_oracle(dev, B, C, Tx, Ty, sim):
    import aligner_amd
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(B * 1000 + Tx)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    t_x[0] = Tx
    temp = 0.0005 if sim == "l2" else 0.11
    want, want_soft = S.soft_attention(k, q, t_x=t_x, temperature=temp, sim=sim)
    got, soft = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev), temperature=temp, sim=sim,
                                           want_soft=True)
    torch.cuda.synchronize()
    assert _cmp(got, want) < TOL
    assert (soft.cpu() - want_soft).abs().max().item() < TOL
    # columns are normalised over the valid text rows
    assert torch.allclose(soft.sum(1).cpu(), torch.ones(B, Ty), atol=1e-4)


@pytest.mark.parametrize("B,C,Tx,Ty,sim,prior", [(2, 80, 300, 700, "l2", False), (1, 80, 500, 1000, "l2", True), (3, 80, 230, 130, "dot", False),
                                                  (2, 128, 260, 257, "l2", False), (1, 256, 150, 400, "l2", True), (8, 80, 500, 1030, "l2", False)])
def test_row_group_form_with_several_waves_per_strip(dev, request, B, C, Tx, Ty, sim, prior):
    """Long text on a small batch: the row-group form gives every 32-frame strip to two or four waves (the row tiles dealt
    round-robin, their running maxima and sums met through LDS) while one wave per strip would leave SIMDs idle.  Against the oracle at 1e-4
    and against the one-wave-per-strip form (`softattn_no_pair`) -- the same logits, the log-sum combined in another order."""
    import aligner_amd
    from aligner_amd import _lib
    from oracle import softattn_oracle as S
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 31 + Tx + Ty)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    t_x[0] = Tx
    pr = torch.rand(B, Tx, Ty, generator=g) if prior else None
    temp = 0.0005 if sim == "l2" else 0.11
    want, _ = S.soft_attention(k, q, t_x=t_x, prior=pr, temperature=temp, sim=sim)
    kw = dict(t_x=t_x.to(dev), prior=None if pr is None else pr.to(dev), temperature=temp, sim=sim)
    got, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), **kw)
    request.addfinalizer(lambda: (lib.aligner_debug_set_option(b"softattn_no_pair", 0), lib.aligner_debug_set_option(b"softattn_split", 0)))
    forced = []
    for sp in (2, 4):                                   # both splits whatever the launch would pick
        assert lib.aligner_debug_set_option(b"softattn_split", sp) == 0
        forced.append(aligner_amd.soft_attention(k.to(dev), q.to(dev), **kw)[0])
    lib.aligner_debug_set_option(b"softattn_split", 0)
    assert lib.aligner_debug_set_option(b"softattn_no_pair", 1) == 0
    one, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), **kw)
    torch.cuda.synchronize()
    assert _cmp(got, want) < TOL
    fin = torch.isfinite(one)
    for t in [got] + forced:
        assert torch.equal(torch.isfinite(t), fin) and (t[fin] - one[fin]).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,C,Tx,Ty", [(8, 80, 200, 1000), (5, 80, 77, 300), (8, 128, 224, 257)])
def test_logp_only_path(dev, B, C, Tx, Ty):
    """No soft output, no prior: the store path the benchmark and the DP pipeline use."""
    import aligner_amd
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(7 * B + Ty)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    t_x[0] = Tx
    want, _ = S.soft_attention(k, q, t_x=t_x)
    got, none = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev))
    torch.cuda.synchronize()
    assert none is None and _cmp(got, want) < TOL


@pytest.mark.parametrize("B,C,Tx,Ty,sim", [(2, 80, 50, 130, "l2"), (3, 80, 200, 333, "l2"), (2, 16, 7, 40, "l2"),
                                           (2, 80, 224, 129, "dot"), (1, 80, 300, 257, "l2"), (1, 200, 130, 64, "l2"),
                                           (2, 128, 500, 260, "dot")])
def test_exact_product_kernel_matches_oracle(dev, request, B, C, Tx, Ty, sim):
    """The fp32-MFMA form of the front end (taken for sharp temperatures; forced here through the debug option)
    on the same shapes as the bf16x3 kernels: log-probs, soft output, ragged text, with and without a prior."""
    import aligner_amd
    from aligner_amd import _lib
    from oracle import softattn_oracle as S
    _lib.check(_lib.load().aligner_debug_set_option(b"softattn_exact", 1))
    request.addfinalizer(lambda: _lib.load().aligner_debug_set_option(b"softattn_exact", 0))
    g = torch.Generator().manual_seed(B * 77 + Tx)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    t_x[0] = Tx
    prior = torch.rand(B, Tx, Ty, generator=g)
    temp = 0.0005 if sim == "l2" else 0.11
    for pr in (None, prior):
        want, want_soft = S.soft_attention(k, q, t_x=t_x, prior=pr, temperature=temp, sim=sim)
        got, soft = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev), temperature=temp, sim=sim,
                                               prior=None if pr is None else pr.to(dev), want_soft=True)
        torch.cuda.synchronize()
        assert _cmp(got, want) < TOL
        assert (soft.cpu() - want_soft).abs().max().item() < TOL


# Row-tile form of the similarity kernel (softattn_rt_kernel: one row group, C = 80 or 128, frames in quads, log-probs
# only): every structural case of it -- one strip, a partial last strip, a partial last workgroup, text that leaves whole
# row tiles masked, batches the XCD-aware map does and does not apply to, both k-step counts, both similarities.
RT_SHAPES = [(1, 80, 200, 4, "l2"), (2, 80, 7, 32, "l2"), (3, 80, 224, 36, "dot"), (2, 80, 33, 64, "l2"),
             (5, 80, 200, 260, "l2"), (8, 80, 129, 512, "l2"), (2, 128, 224, 300, "l2"), (16, 128, 64, 96, "dot"),
             (3, 128, 1, 1000, "l2"), (9, 80, 100, 772, "l2")]


@pytest.mark.parametrize("B,C,Tx,Ty,sim", RT_SHAPES)
def test_row_tile_form(dev, request, B, C, Tx, Ty, sim):
    """Against the oracle at 1e-4 and against the strip-per-wave kernel (debug option `softattn_strips`: the same logits,
    the log-sum-exp combined in another order); bf16 log-probs = the fp32 ones rounded to nearest even, bit for bit."""
    import aligner_amd
    from aligner_amd import _lib
    from oracle import softattn_oracle as S
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 131 + Tx * 7 + Ty)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    t_x[0] = Tx
    if B > 1:
        t_x[1] = 1                                       # one valid row: six row tiles fully masked
    temp = 0.0005 if sim == "l2" else 0.11
    want, _ = S.soft_attention(k, q, t_x=t_x, temperature=temp, sim=sim)
    kw = dict(t_x=t_x.to(dev), temperature=temp, sim=sim)
    got, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), **kw)
    b16, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), logp_dtype=torch.bfloat16, **kw)
    request.addfinalizer(lambda: lib.aligner_debug_set_option(b"softattn_strips", 0))
    assert lib.aligner_debug_set_option(b"softattn_strips", 1) == 0
    strips, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), **kw)
    lib.aligner_debug_set_option(b"softattn_strips", 0)
    torch.cuda.synchronize()
    assert _cmp(got, want) < TOL
    fin = torch.isfinite(strips)
    assert torch.equal(torch.isfinite(got), fin) and (got[fin] - strips[fin]).abs().max().item() < 2e-5
    assert torch.equal(b16.cpu(), got.bfloat16().cpu())


@pytest.mark.parametrize("B,C,Tx,Ty,sim", RT_SHAPES)
def test_row_tile_form_with_a_row_pitch(dev, B, C, Tx, Ty, sim):
    """aligner_softattn_ld: the log-probs at a row pitch of whole 128-byte lines (softattn.pitched_logp, what the pipeline
    hands from the similarity kernel to the search) -- the same bits as the contiguous layout, fp32 and bf16, and not a
    byte of the padding columns (or of the canary pitch behind them) touched."""
    import aligner_amd
    from aligner_amd.softattn import pitched_logp
    g = torch.Generator().manual_seed(B * 17 + Tx + Ty)
    k = torch.randn(B, C, Tx, generator=g).to(dev)
    q = torch.randn(B, C, Ty, generator=g).to(dev)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    temp = 0.0005 if sim == "l2" else 0.11
    for dt in (torch.float32, torch.bfloat16):
        flat, _ = aligner_amd.soft_attention(k, q, t_x=t_x, temperature=temp, sim=sim, logp_dtype=dt)
        out = pitched_logp(B, Tx, Ty, dev, dt)
        ld = out.stride(1)
        assert ld % (128 // out.element_size()) == 0 and ld >= Ty and out.stride(0) == Tx * ld
        whole = torch.as_strided(out, (B, Tx, ld), (Tx * ld, ld, 1))
        whole.fill_(-7.0)
        got, _ = aligner_amd.soft_attention(k, q, t_x=t_x, temperature=temp, sim=sim, out=out)
        torch.cuda.synchronize()
        assert got.data_ptr() == out.data_ptr()
        assert torch.equal(got, flat)
        assert bool((whole[:, :, Ty:] == -7.0).all())
    # forms without a row pitch of their own say so
    prior = torch.rand(B, Tx, Ty, generator=g).to(dev)
    if pitched_logp(B, Tx, Ty, dev).stride(1) != Ty:
        from aligner_amd._lib import AlignerError
        with pytest.raises(AlignerError):
            aligner_amd.soft_attention(k, q, t_x=t_x, prior=prior, out=pitched_logp(B, Tx, Ty, dev))


def test_row_tile_form_is_deterministic_under_uneven_load(dev, request):
    """The compute waves take a strip's normaliser from the loader wave through LDS (polled, with their own merge of the
    published statistics as the fall-back): both roads give the same bits, so the result must not depend on timing.
    The bench shape again and again while a second stream keeps part of the chip busy with launches of other sizes."""
    import aligner_amd
    B, C, Tx, Ty = 64, 80, 200, 1000
    g = torch.Generator().manual_seed(99)
    k = torch.randn(B, C, Tx, generator=g).to(dev)
    q = torch.randn(B, C, Ty, generator=g).to(dev)
    t_x = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    ref, _ = aligner_amd.soft_attention(k, q, t_x=t_x)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    k2, q2 = k[:5, :, :70].contiguous(), q[:5, :, :388].contiguous()
    outs = []
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(1 + it % 3):
                aligner_amd.soft_attention(k2, q2)
        outs.append(aligner_amd.soft_attention(k, q, t_x=t_x)[0])
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.view(torch.int32), ref.view(torch.int32))
    # ... and with a loader that never publishes (every wave takes the fall-back, every strip)
    from aligner_amd import _lib
    lib = _lib.load()
    request.addfinalizer(lambda: lib.aligner_debug_set_option(b"softattn_rt_drop_merge", 0))
    assert lib.aligner_debug_set_option(b"softattn_rt_drop_merge", 1) == 0
    alone, _ = aligner_amd.soft_attention(k, q, t_x=t_x)
    lib.aligner_debug_set_option(b"softattn_rt_drop_merge", 0)
    torch.cuda.synchronize()
    assert torch.equal(alone.view(torch.int32), ref.view(torch.int32))


def _oracle_per_utterance(k, q, t_x, **kw):
    """The oracle evaluated one utterance at a time (its [B,C,Tx,Ty] difference tensor is 4 GB at B = 64)."""
    from oracle import softattn_oracle as S
    return torch.cat([S.soft_attention(k[b:b + 1], q[b:b + 1], t_x=None if t_x is None else t_x[b:b + 1], **kw)[0]
                      for b in range(k.shape[0])])


def test_bench_shape_matches_oracle(dev):
    """BASELINE configs[1] at its full size [64,80,200,1000] -- the shape bench.py times, with bench.py's
    inputs: the similarity kernel's workgroup -> (utterance, frame block) map depends on B (XCD-aware remap),
    so the small-batch cases above do not cover it.  Full-length and ragged text."""
    import aligner_amd
    B, C, Tx, Ty = 64, 80, 200, 1000
    g = torch.Generator().manual_seed(1234)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    full = torch.full((B,), Tx, dtype=torch.int32)
    ragged = torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32)
    for t_x in (full, ragged):
        got, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev))
        torch.cuda.synchronize()
        want = _oracle_per_utterance(k, q, t_x)
        assert _cmp(got, want) < TOL


def test_prior_and_sharp_temperature(dev):
    import aligner_amd
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(5)
    B, C, Tx, Ty = 2, 80, 60, 200
    k = torch.randn(B, C, Tx, generator=g) * 3
    q = torch.randn(B, C, Ty, generator=g) * 3
    prior = torch.rand(B, Tx, Ty, generator=g)
    t_x = torch.tensor([60, 41], dtype=torch.int32)
    for temp in (0.0005, 0.05):
        want, want_soft = S.soft_attention(k, q, t_x=t_x, prior=prior, temperature=temp)
        got, soft = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev), prior=prior.to(dev),
                                               temperature=temp, want_soft=True)
        torch.cuda.synchronize()
        # 1e-4 at the sharp temperature too (|logp| up to ~150 here): above temperature 0.002 the host takes the
        # exact-product kernel (fp32 MFMA) instead of the bf16x3 one -- rule in aligner_softattn_f32
        assert _cmp(got, want) < TOL
        assert (soft.cpu() - want_soft).abs().max().item() < TOL


def test_conv1d_and_full_encoder(dev):
    import aligner_amd
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(11)
    for (B, Ci, Co, T, K, relu) in [(2, 80, 160, 300, 3, True), (2, 160, 80, 300, 1, False), (1, 33, 70, 129, 5, True),
                                    (2, 512, 1024, 50, 3, True)]:
        x = torch.randn(B, Ci, T, generator=g)
        w = torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5
        b = torch.randn(Co, generator=g)
        want = S.conv1d(x, w, b, relu)
        got = aligner_amd.conv1d(x.to(dev), w.to(dev), b.to(dev), relu)
        torch.cuda.synchronize()
        assert (got.cpu() - want).abs().max().item() < 1e-4
    params = aligner_amd.AlignmentEncoderParams.random(64, 80, 80, dev, seed=1)
    text = torch.randn(2, 64, 40, generator=g)
    mel = torch.randn(2, 80, 170, generator=g)
    t_x = torch.tensor([40, 25], dtype=torch.int32)
    got, _ = aligner_amd.alignment_encoder(text.to(dev), mel.to(dev), params, t_x=t_x.to(dev))
    cpu = lambda st: [(w.cpu(), b.cpu()) for w, b in st]  # noqa: E731
    want, _ = S.alignment_encoder(text, mel, cpu(params.key_proj), cpu(params.query_proj), t_x=t_x)
    torch.cuda.synchronize()
    assert _cmp(got, want) < TOL


@pytest.mark.parametrize("B,Ci,Co,T,K,relu", [
    (3, 512, 1024, 200, 3, True),      # the C3 text encoder's wide layer: one 13-tile workgroup per utterance
    (2, 128, 128, 16, 1, False),       # the smallest shape the wide form takes
    (2, 130, 130, 201, 3, True),       # channels not a multiple of 32 / 128, T % 4 != 0 (scalar stores)
    (2, 160, 256, 900, 5, False),      # several frame tiles per utterance, k = 5 (two halo frames)
    (1, 256, 384, 129, 1, True),       # T just over 128: the 13-tile form with most of it empty
    (2, 140, 200, 1000, 3, True),      # 8-tile workgroups (1000 frames: 1024 vs 1040)
    (2, 64, 128, 16, 1, False),        # Cin < 128: the narrow form with 8 waves x 16 channels
    # narrow layers (conv_narrow_kernel: every wave all frames of the tile, activations of all chunks staged at once)
    (3, 80, 160, 900, 3, True),        # mel encoder layer 1: 5 waves x 32 channels
    (3, 160, 80, 900, 1, True),        # mel encoder layer 2: 5 waves x 16 channels
    (2, 80, 80, 333, 1, False),        # T % 4 != 0, T % 16 != 0
    (3, 1024, 80, 200, 1, False),      # text encoder projection: 32 chunks, 2 frame tiles per workgroup
    (1, 33, 70, 129, 5, True),         # ragged everything, k = 5
    (2, 16, 256, 40, 3, False),        # 8 waves x 32 channels, half a chunk of input
])
def test_wide_conv_gemm_form(dev, B, Ci, Co, T, K, relu):
    """csrc/convgemm.hip (activations split + LDS-DMA staging + weight fragments in registers) against the fp32 oracle at
    1e-4, and against the round-3 kernel it replaces on these layers (same bf16x3 products: agreement far inside that)."""
    import aligner_amd
    from aligner_amd import _lib
    from oracle import softattn_oracle as S
    lib = _lib.load()
    assert lib.aligner_conv1d_workspace_bytes(B, Ci, Co, T, K) > 0, "this shape is meant to take the GEMM form"
    g = torch.Generator().manual_seed(B * 1000 + Ci + T)
    x = torch.randn(B, Ci, T, generator=g)
    w = torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5
    b = torch.randn(Co, generator=g)
    want = S.conv1d(x, w, b, relu)
    got = aligner_amd.conv1d(x.to(dev), w.to(dev), b.to(dev), relu)
    torch.cuda.synchronize()
    assert (got.cpu() - want).abs().max().item() < 1e-4
    # the round-3 kernel through the workspace-free entry point, on the same prepared buffer
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    n = lib.aligner_conv1d_prepared_bytes(Co, Ci, K)
    prep = torch.empty(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.aligner_conv1d_prepare_f32(wd.data_ptr(), prep.data_ptr(), n, Co, Ci, K, st))
    y_old = torch.empty(B, Co, T, device=dev)
    _lib.check(lib.aligner_conv1d_prepared_f32(xd.data_ptr(), prep.data_ptr(), bd.data_ptr(), y_old.data_ptr(), B, Ci, Co, T, K,
                                               int(relu), st))
    torch.cuda.synchronize()
    assert (got - y_old).abs().max().item() < 2e-5
    # no bias
    got = aligner_amd.conv1d(xd, wd, None, relu)
    assert (got.cpu() - S.conv1d(x, w, None, relu)).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,T,chans,ks", [
    (3, 333, (80, 160, 80, 80), (3, 1, 1)),        # the mel encoder: waves own (2, 1, 1) output tiles in the three layers
    (2, 200, (48, 80, 80), (5, 1)),                # (1, 1): six waves, the image's last chunk half padding
    (2, 130, (96, 80, 80, 80), (1, 1, 1)),         # (1, 1, 1)
    (2, 90, (64, 160, 160), (3, 1)),               # (2, 2)
    (5, 100, (64, 160, 80), (1, 1)),               # (2, 1)
    (64, 900, (80, 160, 80, 80), (3, 1, 1)),       # BASELINE configs[2]'s mel encoder at its full size
])
def test_trailing_narrow_layers_in_one_kernel(dev, B, T, chans, ks):
    """conv_narrow_fused_kernel: a stack's trailing narrow layers (k = 1 after the first) with the tiles kept in LDS between
    the layers -- against the fp32 oracle stack at 1e-4 and against the same stack as separate kernels (debug option)."""
    from aligner_amd import _lib
    from aligner_amd.softattn import encode
    from oracle import softattn_oracle as S
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, chans[0], T, generator=g)
    stack = [(torch.randn(co, ci, k, generator=g) / (ci * k) ** 0.5, torch.randn(co, generator=g) * 0.1)
             for ci, co, k in zip(chans[:-1], chans[1:], ks)]
    dstack = [(w.to(dev), b.to(dev)) for w, b in stack]
    xd = x.to(dev)
    for rep in range(2):
        got = encode(xd, dstack)
    try:
        assert lib.aligner_debug_set_option(b"conv_no_fuse", 1) == 0
        sep = encode(xd, dstack)
    finally:
        lib.aligner_debug_set_option(b"conv_no_fuse", 0)
    torch.cuda.synchronize()
    assert (got - sep).abs().max().item() < 5e-5
    if B * T <= 4000:                                       # (the full-size case is checked against the separate kernels only)
        want = S.encode(x, stack)
        assert (got.cpu() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("ft", [8, 4, 2])
def test_narrow_conv_every_frame_tile(dev, ft):
    """conv_narrow_kernel's three tile sizes (the launch picks one from the batch size: small test batches would only
    ever see the smallest), forced through the debug option, alone and chained."""
    import aligner_amd
    from aligner_amd import _lib
    from aligner_amd.softattn import encode
    from oracle import softattn_oracle as S
    lib = _lib.load()
    g = torch.Generator().manual_seed(ft)
    try:
        assert lib.aligner_debug_set_option(b"conv_narrow_ft", ft) == 0
        for (B, T, chans, ks) in [(3, 333, (80, 160, 80, 80), (3, 1, 1)), (2, 200, (96, 48), (5,)), (5, 129, (160, 96, 33), (1, 1))]:
            x = torch.randn(B, chans[0], T, generator=g)
            stack = [(torch.randn(co, ci, k, generator=g) / (ci * k) ** 0.5, torch.randn(co, generator=g) * 0.1)
                     for ci, co, k in zip(chans[:-1], chans[1:], ks)]
            want = S.encode(x, stack)
            for rep in range(3):                           # (the round-4 race showed in one run out of a few)
                got = encode(x.to(dev), [(w.to(dev), b.to(dev)) for w, b in stack])
                torch.cuda.synchronize()
                assert (got.cpu() - want).abs().max().item() < 1e-4
    finally:
        lib.aligner_debug_set_option(b"conv_narrow_ft", 0)


@pytest.mark.parametrize("B,T,chans,ks", [
    (3, 200, (512, 1024, 80), (3, 1)),             # the text encoder of SURVEY 7.4
    (3, 900, (80, 160, 80, 80), (3, 1, 1)),        # the mel encoder
    (2, 150, (40, 64, 48, 130, 20), (5, 3, 3, 1)), # consumers with k != 1: fp32 temporary + split pass between the layers
    (2, 77, (160, 300), (1,)),                     # one layer, 300 output channels (wide form, three output tiles)
    (2, 77, (128, 512, 33), (3, 1)),               # 16 chunks into a k = 1 layer (the group-ring form), ragged frame tile, 33 channels
    (5, 333, (128, 640, 160), (1, 1)),             # 20 chunks, 160 output channels (two tiles a wave), T % 32 != 0
    (3, 150, (128, 512, 64, 64, 64, 48), (3, 1, 1, 1, 1)),   # the group-ring layer in the middle of a stack: its epilogue writes
                                                   # the next layer's split image (SPLIT), the last three layers run as one kernel
])
def test_conv_stack_one_call(dev, B, T, chans, ks):
    """aligner_conv_stack_f32 (encode()): the whole stack in one call, layers chained through split images, against the
    fp32 oracle stack at 1e-4 and against the same layers run one by one."""
    import aligner_amd
    from aligner_amd import _lib
    from aligner_amd.softattn import encode
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, chans[0], T, generator=g)
    stack = []
    for ci, co, k in zip(chans[:-1], chans[1:], ks):
        stack.append((torch.randn(co, ci, k, generator=g) / (ci * k) ** 0.5, torch.randn(co, generator=g) * 0.1))
    want = S.encode(x, stack)
    dstack = [(w.to(dev), b.to(dev)) for w, b in stack]
    lib = _lib.load()
    layers = (_lib.ConvLayer * len(stack))(*[_lib.ConvLayer(None, None, w.shape[1], w.shape[0], w.shape[2], 0) for w, _ in stack])
    assert lib.aligner_conv_stack_workspace_bytes(layers, len(stack), B, T) > 0, "this stack is meant to run as one call"
    got = encode(x.to(dev), dstack)
    torch.cuda.synchronize()
    assert tuple(got.shape) == tuple(want.shape)
    assert (got.cpu() - want).abs().max().item() < 1e-4
    y = x.to(dev)
    for n, (w, b) in enumerate(dstack):
        y = aligner_amd.conv1d(y, w, b, relu=(n + 1 < len(dstack)))
    assert (got - y).abs().max().item() < 5e-5
    # a k = 1 layer with many input chunks: chunks in groups through a ring of LDS buffers (conv_narrow_ring_kernel) /
    # all staged at once (conv_narrow_kernel): the same sums in another order; each of them the same bits call after call
    again = encode(x.to(dev), dstack)
    assert torch.equal(again, got)
    assert lib.aligner_debug_set_option(b"conv_no_ring", 1) == 0
    try:
        flat = encode(x.to(dev), dstack)
    finally:
        lib.aligner_debug_set_option(b"conv_no_ring", 0)
    assert (got - flat).abs().max().item() < 5e-5


def test_pipeline_similarity_then_dp(dev):
    """configs[1]: similarity + DP.  The hard path from the HIP log-probs equals the
    oracle DP run on the same log-probs (bit-exact integer path)."""
    import aligner_amd
    from oracle import maxpath_oracle as O
    g = torch.Generator().manual_seed(21)
    B, C, Tx, Ty = 4, 80, 90, 400
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_x = torch.tensor([90, 60, 33, 1], dtype=torch.int32)
    t_y = torch.tensor([400, 311, 200, 17], dtype=torch.int32)
    logp, _ = aligner_amd.soft_attention(k.to(dev), q.to(dev), t_x=t_x.to(dev))
    res = aligner_amd.align(logp, t_x.to(dev), t_y.to(dev), path_dtype=torch.int32)
    torch.cuda.synchronize()
    v = logp.cpu().numpy().copy()
    want = np.zeros(v.shape, np.int32)
    O.maximum_path_c(want, v, t_x.numpy().copy(), t_y.numpy().copy())
    assert np.array_equal(res.path.cpu().numpy(), want)


def test_bf16_log_probs_and_long_form_pipeline(dev):
    """bf16 log-prob output (round to nearest even of the fp32 result, bit for bit) on all three kernel forms,
    and BASELINE config C5's data path at its full size [8, T_text=500, T_mel=4000]: bf16 similarity straight
    into the DP (no conversion pass), int32 path equal to the oracle DP on the same bf16 log-probs."""
    import aligner_amd
    from aligner_amd import _lib
    from oracle import maxpath_oracle as O
    g = torch.Generator().manual_seed(55)
    for (B, C, Tx, Ty, exact) in [(3, 80, 200, 520, False), (2, 80, 300, 264, False), (2, 80, 70, 200, True)]:
        k, q = torch.randn(B, C, Tx, generator=g).to(dev), torch.randn(B, C, Ty, generator=g).to(dev)
        t_x = torch.tensor([Tx] + [Tx // 2] * (B - 1), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().aligner_debug_set_option(b"softattn_exact", int(exact)))
        try:
            f32, _ = aligner_amd.soft_attention(k, q, t_x=t_x)
            b16, _ = aligner_amd.soft_attention(k, q, t_x=t_x, logp_dtype=torch.bfloat16)
        finally:
            _lib.load().aligner_debug_set_option(b"softattn_exact", 0)
        torch.cuda.synchronize()
        assert b16.dtype == torch.bfloat16 and torch.equal(b16.cpu(), f32.bfloat16().cpu()), (B, C, Tx, Ty)
    B, C, Tx, Ty = 8, 80, 500, 4000
    k, q = torch.randn(B, C, Tx, generator=g).to(dev), torch.randn(B, C, Ty, generator=g).to(dev)
    t_x = torch.tensor([500, 480, 333, 250, 500, 77, 412, 499], dtype=torch.int32)
    t_y = torch.tensor([4000, 3999, 2800, 2100, 3504, 700, 3333, 4000], dtype=torch.int32)
    logp, _ = aligner_amd.soft_attention(k, q, t_x=t_x.to(dev), logp_dtype=torch.bfloat16)
    res = aligner_amd.align(logp, t_x.to(dev), t_y.to(dev), path_dtype=torch.int32)
    torch.cuda.synchronize()
    v = logp.float().cpu().numpy().copy()
    want = np.zeros(v.shape, np.int32)
    O.maximum_path_c(want, v, t_x.numpy().copy(), t_y.numpy().copy())
    assert np.array_equal(res.path.cpu().numpy(), want)
    assert np.array_equal(res.durations.cpu().numpy().sum(1), t_y.numpy())


@pytest.mark.parametrize("B", [6, 64])
def test_c3_shape_full_pipeline(dev, B):
    """BASELINE config C3: conv text/mel encoders -> log-probs -> DP on an LJSpeech-shaped batch
    (80-dim mel, ~900 frames, 512-dim text embeddings), at a small batch and at the config's B = 64
    (oracle encoders on the whole batch, its similarity one utterance at a time)."""
    import aligner_amd
    from oracle import maxpath_oracle as O
    from oracle import softattn_oracle as S
    g = torch.Generator().manual_seed(33)
    Ct, Cm, Tx, Ty = 512, 80, 180, 900
    params = aligner_amd.AlignmentEncoderParams.random(Ct, Cm, 80, dev, seed=3)
    text = torch.randn(B, Ct, Tx, generator=g)
    mel = torch.randn(B, Cm, Ty, generator=g)
    t_y = torch.randint(170, Ty + 1, (B,), generator=g, dtype=torch.int32)
    t_x = torch.minimum(torch.randint(20, Tx + 1, (B,), generator=g, dtype=torch.int32), t_y // 4)
    t_x[0], t_y[0] = Tx, Ty
    logp, _ = aligner_amd.alignment_encoder(text.to(dev), mel.to(dev), params, t_x=t_x.to(dev))
    res = aligner_amd.align(logp, t_x.to(dev), t_y.to(dev), path_dtype=torch.int32)
    torch.cuda.synchronize()
    cpu = lambda st: [(w.cpu(), b.cpu()) for w, b in st]  # noqa: E731
    want_lp = _oracle_per_utterance(S.encode(text, cpu(params.key_proj)), S.encode(mel, cpu(params.query_proj)), t_x,
                                    temperature=params.temperature)
    assert _cmp(logp, want_lp) < TOL
    v = logp.cpu().numpy().copy()
    want = np.zeros(v.shape, np.int32)
    O.maximum_path_c(want, v, t_x.numpy().copy(), t_y.numpy().copy())
    assert np.array_equal(res.path.cpu().numpy(), want)          # DP on the HIP log-probs: bit-exact
    assert int(res.durations.sum()) == int(t_y.sum())
