"""Fused similarity -> log-softmax -> alignment search (no reference counterpart) -- closed as
"measured, loses" and kept in aligner_amd.experimental for the record; this keeps its parity green.
Log-probabilities against the fp32 oracle at 1e-4 (parity UNPINNED, as for the unfused front end); the path
against the pinned maximum_path oracle run on the log-probabilities the fused kernel itself wrote (bit-exact)."""
import numpy as np
import pytest
import torch

from aligner_amd.experimental import fused_align

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _oracle_logp(k, q, t_x, **kw):
    from oracle import softattn_oracle as S
    return torch.cat([S.soft_attention(k[b:b + 1], q[b:b + 1], t_x=t_x[b:b + 1], **kw)[0] for b in range(k.shape[0])])


@pytest.mark.parametrize("B,C,Tx,Ty,sim", [(2, 80, 50, 130, "l2"), (3, 80, 200, 1000, "l2"), (2, 64, 252, 640, "l2"),
                                           (2, 80, 130, 333, "dot"), (4, 16, 7, 40, "l2"), (2, 80, 64, 2048, "l2"),
                                           (2, 80, 63, 97, "l2"), (2, 33, 127, 500, "l2")])
def test_fused_matches_unfused_semantics(dev, B, C, Tx, Ty, sim):
    import aligner_amd
    from oracle import maxpath_oracle as O
    g = torch.Generator().manual_seed(B * 31 + Tx)
    k = torch.randn(B, C, Tx, generator=g)
    q = torch.randn(B, C, Ty, generator=g)
    t_y = torch.randint(max(Tx, Ty // 2), Ty + 1, (B,), generator=g, dtype=torch.int32)
    t_x = torch.minimum(torch.randint(1, Tx + 1, (B,), generator=g, dtype=torch.int32), t_y)
    t_x[0], t_y[0] = Tx, Ty
    temp = 0.0005 if sim == "l2" else 0.11
    logp, res = fused_align(k.to(dev), q.to(dev), t_x.to(dev), t_y.to(dev), temperature=temp, sim=sim,
                                        path_dtype=torch.int32, want_tok=True)
    torch.cuda.synchronize()
    want = _oracle_logp(k, q, t_x, temperature=temp, sim=sim)
    got = logp.cpu()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin)
    assert (got[fin] - want[fin]).abs().max().item() < 1e-4
    # the DP consumed exactly what was written: the pinned oracle on the kernel's own log-probs
    v = got.numpy().copy()
    v[~np.isfinite(v)] = 0.0                           # rows >= t_x (-inf): never read by the DP
    wantp = np.zeros(v.shape, np.int32)
    O.maximum_path_c(wantp, v, t_x.numpy().copy(), t_y.numpy().copy())
    assert np.array_equal(res.path.cpu().numpy(), wantp)
    assert np.array_equal(res.durations.cpu().numpy(), wantp.sum(2))
    # and the unfused pair on the same inputs gives the same path when its log-probs are the same numbers
    assert aligner_amd.read_status(dev) == 0


def test_fused_without_logp_output(dev):
    import aligner_amd
    g = torch.Generator().manual_seed(3)
    k, q = torch.randn(4, 80, 200, generator=g).to(dev), torch.randn(4, 80, 1000, generator=g).to(dev)
    t_x = torch.tensor([200, 150, 99, 30], dtype=torch.int32, device=dev)
    t_y = torch.tensor([1000, 800, 640, 333], dtype=torch.int32, device=dev)
    _, a = fused_align(k, q, t_x, t_y, path_dtype=torch.int32)
    none, bres = fused_align(k, q, t_x, t_y, want_logp=False, path_dtype=torch.int32)
    torch.cuda.synchronize()
    assert none is None and torch.equal(a.path, bres.path) and torch.equal(a.durations, bres.durations)
