"""TEST INFRASTRUCTURE -- CPU oracle of the forward-sum alignment objective and the path
expanders that sit either side of the hot path.

Parity status: UNPINNED.  None of this is in the reference snapshot (its README only links
the papers, README.md:49-50, and names the OTA branch, README.md:21-25); the functions below
restate the published formulation the build adopted (DESIGN_HISTORY.md 7):

  forward-sum   the total log-likelihood of all monotonic alignments of `logp[Tx,Ty]` --
                the column recurrence of maximum_path_each (reference core.pyx:17-30) with
                log-sum-exp in place of max -- and its gradient, the posterior occupancy
                of every cell (what `-d loss / d logp` equals);
  prior         the beta-binomial alignment prior of the OTA paper;
  regulate      durations -> frame-level expansion of the text encodings (length regulator).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Everything runs in float64 numpy, vectorised over text rows; sizes used in tests finish in seconds.
"""
from __future__ import annotations

import numpy as np


def _logaddexp(a, b):
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.maximum(a, b)
        out = m + np.log1p(np.exp(-np.abs(a - b)))
    return np.where(np.isneginf(m), -np.inf, out)


def forward_sum_one(logp: np.ndarray, tx: int, ty: int):
    """(log Z, posterior[Tx,Ty]) of one utterance; posterior is 0 outside [0,tx)x[0,ty).

    alpha[x,y] = logaddexp(alpha[x,y-1], alpha[x-1,y-1]) + logp[x,y],  alpha[0,0] = logp[0,0]
    beta [x,y] = logaddexp(beta[x,y+1] + logp[x,y+1], beta[x+1,y+1] + logp[x+1,y+1]),  beta[tx-1,ty-1] = 0
    """
    Tx, Ty = logp.shape
    post = np.zeros((Tx, Ty), np.float64)
    if not (1 <= tx <= ty):
        return -np.inf, post
    lp = logp[:tx, :ty].astype(np.float64)
    alpha = np.full((tx, ty), -np.inf)
    alpha[0, 0] = lp[0, 0]
    for y in range(1, ty):
        up = np.concatenate(([-np.inf], alpha[:-1, y - 1]))
        alpha[:, y] = _logaddexp(alpha[:, y - 1], up) + lp[:, y]
    beta = np.full((tx, ty), -np.inf)
    beta[tx - 1, ty - 1] = 0.0
    for y in range(ty - 2, -1, -1):
        g = beta[:, y + 1] + lp[:, y + 1]
        dn = np.concatenate((g[1:], [-np.inf]))
        beta[:, y] = _logaddexp(g, dn)
    logz = alpha[tx - 1, ty - 1]
    with np.errstate(invalid="ignore"):
        pp = np.exp(alpha + beta - logz)
    post[:tx, :ty] = np.where(np.isfinite(pp), pp, 0.0)
    return float(logz), post


def forward_sum(logp: np.ndarray, t_x: np.ndarray, t_y: np.ndarray):
    """loss[B] = -log Z and grad[B,Tx,Ty] = d loss / d logp = -posterior."""
    B = logp.shape[0]
    loss = np.zeros(B, np.float64)
    grad = np.zeros(logp.shape, np.float64)
    for b in range(B):
        lz, post = forward_sum_one(logp[b], int(t_x[b]), int(t_y[b]))
        loss[b] = -lz
        grad[b] = -post
    return loss, grad


def beta_binomial_prior(tx: int, ty: int, scaling: float = 1.0) -> np.ndarray:
    """prior[x, y] = BetaBinomial(n = tx, a = s*(y+1), b = s*(ty-y)).pmf(x), x < tx, y < ty
    (the OTA alignment prior: frame y of ty puts its mass near text position x ~ tx*y/ty)."""
    from scipy.stats import betabinom
    x = np.arange(tx)
    cols = [betabinom(tx, scaling * (y + 1), scaling * (ty - y)).pmf(x) for y in range(ty)]
    return np.stack(cols, axis=1)


def regulate(h: np.ndarray, durations: np.ndarray, Ty: int) -> np.ndarray:
    """Length regulator: out[b,:,y] = h[b,:,x] for the token x that owns frame y, 0 past sum(dur)."""
    B, C, Tx = h.shape
    out = np.zeros((B, C, Ty), h.dtype)
    for b in range(B):
        idx = np.repeat(np.arange(Tx), durations[b])[:Ty]
        out[b, :, :len(idx)] = h[b][:, idx]
    return out


def ctc_forward_sum(scores: np.ndarray, tx, ty, blank_logprob: float = -1.0):
    """The CTC form of the objective -- the published one -- through torch.nn.functional.ctc_loss in float64 on
    the CPU: an EXTERNAL pin (torch's implementation, not this build's maths).  Exactly the OTA paper's public loss:
    a blank column at `blank_logprob` before the text rows, log_softmax over blank + the utterance's text rows per
    frame, CTC loss of the token sequence 1..t_x, reduction "none".  Returns (loss[B], d loss / d scores [B,Tx,Ty]);
    utterances without a labelling (t_x < 1 or t_x > t_y) get loss inf and gradient 0."""
    import torch
    import torch.nn.functional as F
    x = torch.tensor(np.asarray(scores), dtype=torch.float64, requires_grad=True)
    B = x.shape[0]
    losses, tot = [], 0.0
    for b in range(B):
        K, T = int(tx[b]), int(ty[b])
        if not (1 <= K <= T):
            losses.append(float("inf"))
            continue
        lp = torch.cat([torch.full((1, T), float(blank_logprob), dtype=torch.float64), x[b, :K, :T]], dim=0)   # [K+1, T]
        lp = torch.log_softmax(lp, dim=0).t().unsqueeze(1)                                                   # [T, 1, K+1]
        l = F.ctc_loss(lp, torch.arange(1, K + 1).unsqueeze(0), torch.tensor([T]), torch.tensor([K]), blank=0,
                       reduction="none", zero_infinity=False)[0]
        losses.append(float(l.detach()))
        tot = tot + l
    if isinstance(tot, torch.Tensor):
        tot.backward()
        grad = x.grad.numpy()
    else:
        grad = np.zeros(x.shape, np.float64)
    return np.asarray(losses, np.float64), grad
