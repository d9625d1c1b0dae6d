"""oracle/softattn_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU fp32 restatement (plain torch ops) of the soft-attention front end.

PARITY UNPINNED: the OTA / alignment-encoder source is NOT in the reference
snapshot (/root/reference holds only monotonic_align; README.md:21-25,50 point at
another git branch and at arXiv 2108.10447).  This file therefore restates the
published alignment-learning formulation as specified in SURVEY.md 7.4 -- the
build's own spec -- and the 1e-4 fp32 criterion of BASELINE.json is evaluated
against it.  Only tests/, __graft_entry__.smoke() and bench.py may import it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def conv1d(x, w, b=None, relu=False):
    y = F.conv1d(x.float(), w.float(), None if b is None else b.float(), padding=w.shape[-1] // 2)
    return torch.relu(y) if relu else y


def encode(x, stack):
    for n, (w, b) in enumerate(stack):
        x = conv1d(x, w, b, relu=(n + 1 < len(stack)))
    return x


def soft_attention(keys_enc, queries_enc, t_x=None, prior=None, temperature=0.0005, sim="l2"):
    """keys_enc [B,C,Tx], queries_enc [B,C,Ty] -> (logp [B,Tx,Ty], soft [B,Tx,Ty])."""
    k = keys_enc.float()
    q = queries_enc.float()
    if sim == "l2":
        # sum_c (q[c,j] - k[c,i])^2, laid out [B,Tx,Ty]
        d = ((q[:, :, None, :] - k[:, :, :, None]) ** 2).sum(1)
        logit = -temperature * d
    else:
        logit = temperature * torch.einsum("bci,bcj->bij", k, q)
    B, Tx, Ty = logit.shape
    if t_x is not None:
        rows = torch.arange(Tx)[None, :, None] >= t_x.to(torch.long)[:, None, None]
        logit = logit.masked_fill(rows, float("-inf"))
    logp = torch.log_softmax(logit, dim=1)
    if prior is not None:
        logp = logp + torch.log(prior.float() + 1e-8)
    soft = torch.softmax(logp, dim=1)
    return logp, soft


def alignment_encoder(text_emb, mel, key_proj, query_proj, t_x=None, prior=None, temperature=0.0005):
    k = encode(text_emb, key_proj)
    q = encode(mel, query_proj)
    return soft_attention(k, q, t_x=t_x, prior=prior, temperature=temperature)
