"""The steps either side of the alignment path, on the HIP path (build-defined
specs -- the reference snapshot only links the OTA paper, README.md:21-25,50):

    forward_sum(logp, t_x, t_y)          -log-likelihood of all monotonic alignments (+ gradient)
    beta_binomial_prior(t_x, t_y, ...)   the alignment prior soft_attention() can add
    regulate(h, durations, T_mel)        length regulator: expand text encodings to frames

All arithmetic runs in libaligner_amd.so; torch only owns the buffers.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from .softattn import _chk, _stream

_fs_workspaces = _lib.StreamWorkspaces(zero=False, slack=1.0)


def _fs_workspace(device, nbytes: int) -> torch.Tensor:
    return _fs_workspaces.get(device, nbytes)          # one per (device, stream)


def _lengths(t: torch.Tensor, name: str, B: int, dev) -> torch.Tensor:
    t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.int32).contiguous()
    if t.shape != (B,):
        raise ValueError(f"{name} must have shape [{B}]")
    return t


def forward_sum(logp: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, want_grad: bool = True,
                blank_logprob: Optional[float] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """loss[B] = -log sum over monotonic alignments of prod_y exp(logp[b, x(y), y]) and, if
    `want_grad`, d loss / d logp [B,T_text,T_mel] (= minus the posterior occupancy of each cell).

    blank_logprob (e.g. -1.0): the CTC form the OTA paper's code trains with instead -- a blank column at that
    log-prob before the text, every frame renormalised over blank + text, loss = CTC loss of the tokens 1..t_x;
    equal to torch.nn.functional.ctc_loss on log_softmax(pad(logp)), reduction "none" (tests/test_objective.py)."""
    _lib.require_gpu()
    lp = _chk(logp, "logp")
    if lp.dim() != 3:
        raise ValueError("logp must be [B, T_text, T_mel]")
    B, Tx, Ty = lp.shape
    dev = lp.device
    tx = _lengths(t_x, "t_x", B, dev)
    ty = _lengths(t_y, "t_y", B, dev)
    lib = _lib.load()
    ctc = blank_logprob is not None
    nws = (lib.aligner_forward_sum_ctc_workspace_bytes if ctc else lib.aligner_forward_sum_workspace_bytes)(B, Tx, Ty)
    if nws == 0 and B > 0:
        raise ValueError(f"unsupported shape B={B} T_text={Tx} T_mel={Ty} (T_text <= {1023 if ctc else 1024})")
    ws = _fs_workspace(dev, nws)
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev) if want_grad else None
    if ctc:
        with torch.cuda.device(dev):
            _lib.check(lib.aligner_forward_sum_ctc_f32(lp.data_ptr(), tx.data_ptr(), ty.data_ptr(), float(blank_logprob),
                                                       loss.data_ptr(), None if grad is None else grad.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), B, Tx, Ty, _stream(dev)))
        return loss, grad
    with torch.cuda.device(dev):
        _lib.check(lib.aligner_forward_sum_f32(lp.data_ptr(), tx.data_ptr(), ty.data_ptr(), loss.data_ptr(),
                                               None if grad is None else grad.data_ptr(), ws.data_ptr(), ws.numel(),
                                               B, Tx, Ty, _stream(dev)))
    return loss, grad


class _ForwardSumLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, t_x, t_y, blank_logprob):
        need = ctx.needs_input_grad[0]
        loss, grad = forward_sum(logp.detach(), t_x, t_y, want_grad=need, blank_logprob=blank_logprob)
        ctx.save_for_backward(grad if need else None)
        ctx.in_dtype = logp.dtype                   # (the kernels compute in fp32; a bf16 / fp16 logp gets its gradient back in its dtype)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (grad,) = ctx.saved_tensors
        if grad is None or g_loss is None:
            return None, None, None, None
        return (grad * g_loss.to(grad.dtype).view(-1, 1, 1)).to(ctx.in_dtype), None, None, None


def forward_sum_loss(logp: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, blank_logprob: Optional[float] = -1.0,
                     reduction: str = "mean", zero_infinity: bool = False, length_normalize: bool = False) -> torch.Tensor:
    """The OTA aligner's ForwardSumLoss as an autograd function on the GPU kernels: forward_sum() with its gradient
    attached, so that `forward_sum_loss(logp, t_x, t_y).backward()` reaches whatever differentiable torch code produced
    `logp` (this package's `soft_attention()` kernel is forward-only: SURVEY 8 scopes the front end's forward pass).  One launch pair computes loss AND gradient in the forward pass (both sweeps side by side); backward()
    only scales.  blank_logprob = -1.0: the published CTC form (None: the plain monotonic form).  reduction: "mean" (a plain
    mean over the batch), "sum" or "none".  The two switches of torch.nn.CTCLoss that published aligner code is usually
    written with (neither the reference snapshot nor SNIPPETS.md holds that code: parity unpinned) are here as options, off
    by default: zero_infinity -- an infeasible utterance (t_x > t_y: loss +inf) contributes 0 and no gradient instead of
    making the batch loss +inf; length_normalize -- each utterance's loss is divided by its t_x (CTCLoss's own "mean")."""
    loss = _ForwardSumLoss.apply(logp, t_x, t_y, blank_logprob)
    if zero_infinity:
        loss = torch.where(torch.isinf(loss), torch.zeros_like(loss), loss)
    if length_normalize:
        loss = loss / torch.as_tensor(t_x).to(device=loss.device, dtype=loss.dtype).clamp_min(1)
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise ValueError("reduction must be 'mean', 'sum' or 'none'")


def beta_binomial_prior(t_x: torch.Tensor, t_y: torch.Tensor, T_text: int, T_mel: int, scaling: float = 1.0
                        ) -> torch.Tensor:
    """prior[B,T_text,T_mel]: BetaBinomial(n=t_x, a=s*(y+1), b=s*(t_y-y)).pmf(x); 0 in the padding."""
    _lib.require_gpu()
    t_x = torch.as_tensor(t_x)
    if not t_x.is_cuda:
        raise ValueError("t_x must be a GPU tensor")
    dev = t_x.device
    B = t_x.shape[0]
    tx = _lengths(t_x, "t_x", B, dev)
    ty = _lengths(t_y, "t_y", B, dev)
    out = torch.empty((B, T_text, T_mel), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().aligner_beta_binomial_prior_f32(tx.data_ptr(), ty.data_ptr(), out.data_ptr(), B, T_text,
                                                               T_mel, float(scaling), _stream(dev)))
    return out


def regulate(h: torch.Tensor, durations: torch.Tensor, T_mel: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(out[B,C,T_mel], tok[B,T_mel]): out[b,:,y] = h[b,:,tok[b,y]]; frames past sum(durations[b]) are 0 / -1."""
    _lib.require_gpu()
    hh = _chk(h, "h")
    B, C, Tx = hh.shape
    dev = hh.device
    dur = torch.as_tensor(durations).detach().to(device=dev, dtype=torch.int32).contiguous()
    if dur.shape != (B, Tx):
        raise ValueError("durations must be [B, T_text]")
    out = torch.empty((B, C, T_mel), dtype=torch.float32, device=dev)
    tok = torch.empty((B, T_mel), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().aligner_regulate_f32(hh.data_ptr(), dur.data_ptr(), out.data_ptr(), tok.data_ptr(),
                                                    B, C, Tx, T_mel, _stream(dev)))
    return out, tok
