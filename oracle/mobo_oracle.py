"""oracle/mobo_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU float64 restatement of the MoBoAligner monotonic boundary search (BASELINE config 5,
SURVEY.md section 8 rows a7 / f3).

PARITY UNPINNED: the MoBoAligner code is on another git branch of the reference and is not in the
snapshot -- /root/reference/README.md:9-13 only names the branch, README.md:49 links the paper
(Li et al., "MoBoAligner: a Neural Alignment Model for Non-autoregressive TTS with Monotonic Boundary
Search", Interspeech 2020).  This file restates the paper's boundary formulation as this build reads
it, with the maximum-duration window the reference's README names as that branch's limitation
(README.md:13); every constant and convention below is this build's choice, and the HIP path is
checked against THIS file, which in turn is pinned only to brute-force enumeration of all boundary
sequences on tiny shapes (tests/test_mobo.py).  Only tests/, __graft_entry__.smoke() and bench.py
may import it.

Model.  Tokens i = 0..I-1, frames y = 0..J-1, energies e[i, y] (any real numbers; the similarity /
log-likelihood matrix in the alignment layout [T_text, T_mel]).  A segmentation is a boundary
sequence 0 = b_{-1} < b_0 < ... < b_{I-1} = J: token i owns frames [b_{i-1}, b_i), its duration
d_i = b_i - b_{i-1} is between 1 and D (the maximum-duration window).  Boundaries are searched left
to right; given the previous boundary k, the right boundary of token i is drawn among the feasible
positions with probability proportional to the exponentiated energy of the token at its LAST frame:

    P(b_i = j | b_{i-1} = k) = exp(e[i, j-1]) / sum_{m in A_i(k)} exp(e[i, m-1])
    A_i(k) = { m : k < m <= k + D,  lo_i <= m <= hi_i }
    hi_i = J - (I-1-i)                  (every later token keeps at least one frame)
    lo_i = max(i+1, J - (I-1-i)*D)      (the later tokens can still reach J)

so every row of the chain is a proper distribution and every sequence with positive probability
ends at b_{I-1} = J.  Quantities (all in natural-log units):

    log_alpha[i, j-1] = log P(b_i = j)                                    forward variable
    gamma[i, y]       = P(b_{i-1} <= y < b_i) = cdf_{i-1}(y) - cdf_i(y)   soft alignment, columns sum to 1
    boundaries[i]     = b_i of the MAP sequence (max-product search; ties: the shortest token,
                        i.e. the LARGEST previous boundary), durations = differences
    map_score         = log-probability of that sequence
"""
from __future__ import annotations

import itertools

import numpy as np

NEG = -np.inf


def _bounds(I: int, J: int, D: int, i: int):
    hi = J - (I - 1 - i)
    lo = max(i + 1, J - (I - 1 - i) * D)
    return lo, hi


def feasible(I: int, J: int, D: int) -> bool:
    return 1 <= I <= J <= I * D


def _logsumexp(a):
    a = np.asarray(a, np.float64)
    m = a.max() if a.size else NEG
    if not np.isfinite(m):
        return m
    return m + np.log(np.exp(a - m).sum())


def boundary_search(e: np.ndarray, D: int):
    """e [I, J] float -> dict(log_alpha [I,J], gamma [I,J], boundaries [I], durations [I], map_score).

    Plain loops over tokens and previous boundaries; keep sizes small (I*J*D <= ~1e7)."""
    e = np.asarray(e, np.float64)
    I, J = e.shape
    if not feasible(I, J, D):
        raise ValueError(f"infeasible: need I <= J <= I*D (I={I}, J={J}, D={D})")
    la_prev = np.full(J + 1, NEG)           # log P(b_{i-1} = k), k = 0..J
    la_prev[0] = 0.0
    de_prev = la_prev.copy()                # max-product twin
    log_alpha = np.full((I, J), NEG)
    back = np.zeros((I, J + 1), np.int64)
    cdf_prev = np.ones(J)                   # P(b_{-1} <= y) = 1
    gamma = np.zeros((I, J))
    for i in range(I):
        lo, hi = _bounds(I, J, D, i)
        s = np.full(J + 1, NEG)
        s[1:] = e[i]                        # s[j] = e[i, j-1]: energy at the last frame of a token ending at j
        # normaliser per previous boundary k
        L = np.full(J + 1, NEG)
        for k in range(J):
            a, b = max(k + 1, lo), min(k + D, hi)
            if a <= b and (np.isfinite(la_prev[k]) or np.isfinite(de_prev[k])):
                L[k] = _logsumexp(s[a:b + 1])
        la = np.full(J + 1, NEG)
        de = np.full(J + 1, NEG)
        for j in range(lo, hi + 1):
            ks = np.arange(max(0, j - D), j)
            ok = np.isfinite(L[ks])
            if not ok.any():
                continue
            with np.errstate(invalid="ignore"):
                u = np.where(ok, la_prev[ks] - L[ks], NEG)
                v = np.where(ok, de_prev[ks] - L[ks], NEG)
            la[j] = s[j] + _logsumexp(u)
            vm = v.max()
            if np.isfinite(vm):
                kbest = ks[np.nonzero(v == vm)[0].max()]           # ties: the largest previous boundary
                de[j] = s[j] + vm
                back[i, j] = kbest
        log_alpha[i] = la[1:]
        cdf = np.cumsum(np.exp(la[1:]))                            # P(b_i <= y + 1)... index y: b_i <= y+1
        # gamma[i, y] = P(b_{i-1} <= y) - P(b_i <= y)
        cdf_i = np.concatenate([[0.0], cdf[:-1]])                  # P(b_i <= y), y = 0..J-1
        gamma[i] = cdf_prev - cdf_i
        cdf_prev = cdf_i
        la_prev, de_prev = la, de
    bnd = np.zeros(I, np.int64)
    j = J
    for i in range(I - 1, -1, -1):
        bnd[i] = j
        j = back[i, j]
    dur = np.diff(np.concatenate([[0], bnd]))
    return dict(log_alpha=log_alpha, gamma=gamma, boundaries=bnd.astype(np.int32),
                durations=dur.astype(np.int32), map_score=float(de_prev[J]))


def _lse_rows(w):
    """logsumexp along the last axis of a 2-D array of windows (-inf for an all -inf window)."""
    m = w.max(axis=1)
    ok = np.isfinite(m)
    out = np.full(w.shape[0], NEG)
    if ok.any():
        with np.errstate(invalid="ignore"):
            out[ok] = m[ok] + np.log(np.exp(w[ok] - m[ok, None]).sum(axis=1))
    return out


def boundary_search_fast(e: np.ndarray, D: int):
    """The same quantities as boundary_search() with every token row vectorised over the positions (numpy
    sliding windows): affordable at the full BASELINE config-5 size [500, 4000].  tests/test_mobo.py checks it
    against the plain loops above on moderate shapes before trusting it at sizes the loops cannot reach."""
    from numpy.lib.stride_tricks import sliding_window_view as swv
    e = np.asarray(e, np.float64)
    I, J = e.shape
    if not feasible(I, J, D):
        raise ValueError(f"infeasible: need I <= J <= I*D (I={I}, J={J}, D={D})")
    la_prev = np.full(J + 1, NEG)
    la_prev[0] = 0.0
    de_prev = la_prev.copy()
    log_alpha = np.full((I, J), NEG)
    back = np.zeros((I, J + 1), np.int64)
    cdf_prev = np.ones(J)
    gamma = np.zeros((I, J))
    pos = np.arange(J + 1)
    for i in range(I):
        lo, hi = _bounds(I, J, D, i)
        s = np.full(J + 1, NEG)
        s[1:] = e[i]
        sm = np.where((pos >= lo) & (pos <= hi), s, NEG)
        # L[k] = lse sm[k+1 .. k+D], k = 0..J-1
        spad = np.concatenate([sm[1:], np.full(D, NEG)])
        L = np.full(J + 1, NEG)
        L[:J] = _lse_rows(swv(spad, D)[:J])
        used = np.isfinite(la_prev) | np.isfinite(de_prev)
        L = np.where(used, L, NEG)
        okL = np.isfinite(L)
        with np.errstate(invalid="ignore"):
            u = np.where(okL, la_prev - L, NEG)
            v = np.where(okL, de_prev - L, NEG)
        u = np.where(np.isnan(u), NEG, u)
        v = np.where(np.isnan(v), NEG, v)
        # windows k = j-D .. j-1 for j = 0..J  (k < 0: -inf)
        uw = swv(np.concatenate([np.full(D, NEG), u]), D)[:J + 1]
        vw = swv(np.concatenate([np.full(D, NEG), v]), D)[:J + 1]
        feas = (pos >= lo) & (pos <= hi)
        la = np.full(J + 1, NEG)
        de = np.full(J + 1, NEG)
        idx = np.nonzero(feas)[0]
        la[idx] = s[idx] + _lse_rows(uw[idx])
        vmax = vw[idx].max(axis=1)
        # ties: the largest previous boundary = the LAST maximum of the window
        last = D - 1 - np.argmax(vw[idx][:, ::-1], axis=1)
        okv = np.isfinite(vmax)
        de[idx[okv]] = s[idx[okv]] + vmax[okv]
        back[i, idx[okv]] = idx[okv] - D + last[okv]
        la = np.where(np.isnan(la), NEG, la)
        log_alpha[i] = la[1:]
        cdf = np.cumsum(np.exp(la[1:]))
        cdf_i = np.concatenate([[0.0], cdf[:-1]])
        gamma[i] = cdf_prev - cdf_i
        cdf_prev = cdf_i
        la_prev, de_prev = la, de
    bnd = np.zeros(I, np.int64)
    j = J
    for i in range(I - 1, -1, -1):
        bnd[i] = j
        j = back[i, j]
    dur = np.diff(np.concatenate([[0], bnd]))
    return dict(log_alpha=log_alpha, gamma=gamma, boundaries=bnd.astype(np.int32),
                durations=dur.astype(np.int32), map_score=float(de_prev[J]))


def sequence_log_prob(e: np.ndarray, D: int, boundaries) -> float:
    """log-probability of one boundary sequence under the chain (-inf if it violates a constraint)."""
    e = np.asarray(e, np.float64)
    I, J = e.shape
    k, lp = 0, 0.0
    for i in range(I):
        j = int(boundaries[i])
        lo, hi = _bounds(I, J, D, i)
        a, b = max(k + 1, lo), min(k + D, hi)
        if not (a <= j <= b):
            return NEG
        lp += e[i, j - 1] - _logsumexp(e[i, a - 1:b])
        k = j
    return lp if k == J else NEG


def brute_force(e: np.ndarray, D: int):
    """Enumerate every boundary sequence (tiny shapes only): the pin of boundary_search."""
    e = np.asarray(e, np.float64)
    I, J = e.shape
    log_alpha = np.full((I, J), NEG)
    gamma = np.zeros((I, J))
    best, best_b = NEG, None
    for durs in itertools.product(range(1, D + 1), repeat=I):
        if sum(durs) != J:
            continue
        b = np.cumsum(durs)
        lp = sequence_log_prob(e, D, b)
        if not np.isfinite(lp):
            continue
        for i in range(I):
            log_alpha[i, b[i] - 1] = np.logaddexp(log_alpha[i, b[i] - 1], lp)
            gamma[i, (b[i - 1] if i else 0):b[i]] += np.exp(lp)
        # ties: prefer the lexicographically larger boundary sequence seen from the LAST token backwards,
        # which is what "largest previous boundary first" in the backtrack produces
        if lp > best or (lp == best and tuple(b[::-1]) > tuple(best_b[::-1])):
            best, best_b = lp, b
    return dict(log_alpha=log_alpha, gamma=gamma, boundaries=np.asarray(best_b, np.int32), map_score=float(best))


# ------------------------------------------------------------------------------------------------------------
# The gradient of the search (round 3): what a training step needs behind log_alpha / gamma.
#
# With u_i(k) = la_{i-1}(k) - L_i(k) and q_i(j) = e_i(j) - la_i(j) = -logsumexp_{k in [j-D, j)} u_i(k), the
# posterior of the previous boundary is r_i(k | j) = exp(u_i(k) + q_i(j)) (it sums to one over k), and for a loss
# with the direct cotangent G_i(j) = dl/dla_i(j) the TOTAL derivative Z_i(j) = dl/dla_i(j) follows the backward
# ("beta") recursion over the token rows
#     Y_i(k)   = sum_{j in (k, k+D]} Z_i(j) r_i(k | j)            mass that reaches row i from position k
#     Z_{i-1}  = G_{i-1} + Y_i,            Z_{I-1} = G_{I-1}
#     dl/de[i, m-1] = Z_i(m) - sum_{k in [m-D, m)} exp(e_i(m) - L_i(k)) Y_i(k)
# (the chain is locally normalised, so the classical beta variable is identically one and gamma needs no backward
# pass; THIS recursion is the adjoint of the alpha pass).  gamma is linear in alpha = exp(la):
#     dgamma_i(y)/dalpha_i(j) = -[j <= y],   dgamma_{i+1}(y)/dalpha_i(j) = +[j <= y],
# so a cotangent on gamma enters G as alpha_i(j) * sum_{y >= j} (Gamma_{i+1}(y) - Gamma_i(y)).
# Cotangents at entries whose log_alpha is -inf are ignored.
# ------------------------------------------------------------------------------------------------------------
def _forward_rows(e: np.ndarray, D: int):
    """la[i+1, j] = log P(b_i = j) (row 0: the start), L[i, k]; float64, -inf = log 0."""
    from numpy.lib.stride_tricks import sliding_window_view as swv
    I, J = e.shape
    la = np.full((I + 1, J + 1), NEG)
    la[0, 0] = 0.0
    Ls = np.full((I, J + 1), NEG)
    pos = np.arange(J + 1)
    for i in range(I):
        lo, hi = _bounds(I, J, D, i)
        s = np.full(J + 1, NEG)
        s[1:] = e[i]
        feas = (pos >= lo) & (pos <= hi)
        sm = np.where(feas, s, NEG)
        L = np.full(J + 1, NEG)
        L[:J] = _lse_rows(swv(np.concatenate([sm[1:], np.full(D, NEG)]), D)[:J])
        Ls[i] = L
        with np.errstate(invalid="ignore"):
            u = np.where(np.isfinite(L) & np.isfinite(la[i]), la[i] - L, NEG)
        uw = swv(np.concatenate([np.full(D, NEG), u]), D)[:J + 1]
        idx = np.nonzero(feas)[0]
        row = np.full(J + 1, NEG)
        with np.errstate(invalid="ignore"):
            row[idx] = s[idx] + _lse_rows(uw[idx])
        la[i + 1] = np.where(np.isnan(row), NEG, row)
    return la, Ls


def boundary_search_backward(e: np.ndarray, D: int, grad_log_alpha=None, grad_gamma=None) -> np.ndarray:
    """dl/de [I, J] for l = <grad_log_alpha, log_alpha> + <grad_gamma, gamma> (either may be None), float64."""
    from numpy.lib.stride_tricks import sliding_window_view as swv
    e = np.asarray(e, np.float64)
    I, J = e.shape
    if not feasible(I, J, D):
        raise ValueError(f"infeasible: need I <= J <= I*D (I={I}, J={J}, D={D})")
    la, Ls = _forward_rows(e, D)
    G = np.zeros((I, J + 1))
    if grad_log_alpha is not None:
        g = np.asarray(grad_log_alpha, np.float64)
        G[:, 1:] += np.where(np.isfinite(la[1:, 1:]), g, 0.0)
    if grad_gamma is not None:
        gg = np.concatenate([np.asarray(grad_gamma, np.float64), np.zeros((1, J))])
        for i in range(I):
            suf = np.cumsum((gg[i + 1] - gg[i])[::-1])[::-1]           # sum over y >= j, j = 0..J-1
            S = np.concatenate([suf, [0.0]])
            with np.errstate(over="ignore"):
                G[i] += np.where(np.isfinite(la[i + 1]), np.exp(la[i + 1]) * S, 0.0)
    grad = np.zeros((I, J))
    pos = np.arange(J + 1)
    Y = np.zeros(J + 1)
    for i in range(I - 1, -1, -1):
        lo, hi = _bounds(I, J, D, i)
        s = np.full(J + 1, NEG)
        s[1:] = e[i]
        feas = (pos >= lo) & (pos <= hi) & np.isfinite(s)
        Z = G[i] + Y
        with np.errstate(invalid="ignore"):
            q = np.where(feas & np.isfinite(la[i + 1]), s - la[i + 1], NEG)
            u = np.where(np.isfinite(Ls[i]) & np.isfinite(la[i]), la[i] - Ls[i], NEG)
        q = np.where(np.isnan(q), NEG, q)
        # Y(k) = sum_{j = k+1 .. k+D} Z(j) exp(u(k) + q(j))
        Zw = swv(np.concatenate([Z[1:], np.zeros(D)]), D)[:J + 1]
        qw = swv(np.concatenate([q[1:], np.full(D, NEG)]), D)[:J + 1]
        with np.errstate(invalid="ignore", over="ignore"):
            ex = u[:, None] + qw
            Y = np.where(np.isfinite(ex), Zw * np.exp(np.where(np.isfinite(ex), ex, 0.0)), 0.0).sum(axis=1)
        # grad(m) = Z(m) - sum_{k = m-D .. m-1} exp(s(m) - L(k)) Y(k)
        Lw = swv(np.concatenate([np.full(D, NEG), Ls[i]]), D)[:J + 1]
        Yw = swv(np.concatenate([np.zeros(D), Y]), D)[:J + 1]
        with np.errstate(invalid="ignore", over="ignore"):
            ex = s[:, None] - Lw
            ok = np.isfinite(ex) & feas[:, None]
            flow = np.where(ok, np.exp(np.where(ok, ex, 0.0)) * Yw, 0.0).sum(axis=1)
        grad[i] = np.where(feas, Z - flow, 0.0)[1:]
    return grad


def boundary_search_torch(e, D: int):
    """Differentiable float64 torch restatement of log_alpha / gamma (finite energies only): the pin of
    boundary_search_backward through autograd.  "log 0" is -1e30 inside; returns (log_alpha with -inf, gamma)."""
    import torch
    I, J = e.shape
    NEGF = -1e30
    dt = torch.float64
    e = e.to(dt)
    pos = torch.arange(J + 1)
    la_prev = torch.full((J + 1,), NEGF, dtype=dt)
    la_prev[0] = 0.0
    rows = []
    for i in range(I):
        lo, hi = _bounds(I, J, D, i)
        s = torch.cat([torch.full((1,), NEGF, dtype=dt), e[i]])
        feas = (pos >= lo) & (pos <= hi)
        sm = torch.where(feas, s, torch.full_like(s, NEGF))
        spad = torch.cat([sm[1:], torch.full((D,), NEGF, dtype=dt)])
        L = torch.cat([torch.logsumexp(spad.unfold(0, D, 1)[:J], dim=1), torch.full((1,), NEGF, dtype=dt)])
        live = (L > -1e29) & (la_prev > -1e29)
        u = torch.where(live, la_prev - L, torch.full_like(L, NEGF))
        uw = torch.cat([torch.full((D,), NEGF, dtype=dt), u]).unfold(0, D, 1)[:J + 1]
        lse = torch.logsumexp(uw, dim=1)
        la = torch.where(feas & (lse > -1e29), s + lse, torch.full_like(s, NEGF))
        rows.append(la[1:])
        la_prev = la
    log_alpha = torch.stack(rows)
    alive = log_alpha > -1e29
    alpha = torch.where(alive, torch.exp(torch.where(alive, log_alpha, torch.zeros_like(log_alpha))),
                        torch.zeros_like(log_alpha))
    cdf = torch.cumsum(alpha, dim=1) - alpha                            # P(b_i <= y), y = 0..J-1
    prev = torch.cat([torch.ones((1, J), dtype=dt), cdf[:-1]])
    gamma = prev - cdf
    return torch.where(alive, log_alpha, torch.full_like(log_alpha, float("-inf"))), gamma, alive
