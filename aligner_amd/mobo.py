"""MoBoAligner monotonic boundary search on the HIP path (BASELINE config 5).

Build-defined spec -- the reference snapshot only names the branch and links the paper (README.md:9-13,49):
boundaries 0 = b_-1 < b_0 < ... < b_{t_x-1} = t_y, token durations 1..max_duration,
P(b_i = j | b_{i-1} = k) = softmax over the feasible j in (k, k + max_duration] of energies[b, i, j-1].
All arithmetic runs in libaligner_amd.so (csrc/mobo.hip); torch only owns the buffers.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}
_workspaces = _lib.StreamWorkspaces(zero=True)


class BoundarySearch(NamedTuple):
    boundaries: Optional[torch.Tensor]    # [B,Tx] int32: end boundary b_i of every token (MAP sequence); None without want_map
    durations: Optional[torch.Tensor]     # [B,Tx] int32
    map_score: Optional[torch.Tensor]     # [B] fp32: log-probability of that sequence
    log_alpha: Optional[torch.Tensor]     # [B,Tx,Ty] fp32: log P(b_i = j) at [i, j-1]
    gamma: Optional[torch.Tensor]         # [B,Tx,Ty] fp32: soft alignment P(b_{i-1} <= y < b_i)


def boundary_search(energies: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, max_duration: int,
                    want_log_alpha: bool = False, want_gamma: bool = False, want_map: bool = True) -> BoundarySearch:
    """energies [B,T_text,T_mel] (fp32 / bf16 / fp16, GPU), lengths [B].  Asynchronous on the current stream.

    The library runs only the chain that is asked for: the max-product one (MAP boundaries / durations / score), the
    sum-product one (log_alpha, gamma -- `want_map=False`: a training step's forward pass), or both in one kernel."""
    _lib.require_gpu()
    if energies.dim() != 3 or not energies.is_cuda:
        raise ValueError("energies must be a GPU tensor [B, T_text, T_mel]")
    e = energies.detach()
    if e.dtype not in _DT:
        e = e.float()
    e = e.contiguous()
    B, Tx, Ty = e.shape
    dev = e.device
    tx = torch.as_tensor(t_x).detach().to(device=dev, dtype=torch.int32).contiguous()
    ty = torch.as_tensor(t_y).detach().to(device=dev, dtype=torch.int32).contiguous()
    if tx.shape != (B,) or ty.shape != (B,):
        raise ValueError("t_x / t_y must have one entry per utterance")
    want_log_alpha = want_log_alpha or want_gamma
    if not want_map and not want_log_alpha:
        raise ValueError("boundary_search: nothing asked for (want_map, want_log_alpha and want_gamma are all False)")
    bnd = torch.empty((B, Tx), dtype=torch.int32, device=dev) if want_map else None
    dur = torch.empty((B, Tx), dtype=torch.int32, device=dev) if want_map else None
    score = torch.empty((B,), dtype=torch.float32, device=dev) if want_map else None
    la = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev) if want_log_alpha else None
    ga = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev) if want_gamma else None
    lib = _lib.load()
    if B > 0:
        with torch.cuda.device(dev):
            need = lib.aligner_boundary_search_workspace_bytes_ex(B, Tx, Ty, int(max_duration))
            if need == 0:                   # shape outside the kernels' limits: let the call say why
                need = 256
            ws = _workspaces.get(dev, need)
            _lib.check(lib.aligner_boundary_search(
                e.data_ptr(), _DT[e.dtype], tx.data_ptr(), ty.data_ptr(), int(max_duration),
                None if bnd is None else bnd.data_ptr(), None if dur is None else dur.data_ptr(),
                None if score is None else score.data_ptr(), None if la is None else la.data_ptr(),
                None if ga is None else ga.data_ptr(), ws.data_ptr(), ws.numel(), B, Tx, Ty,
                torch.cuda.current_stream(dev).cuda_stream))
    return BoundarySearch(bnd, dur, score, la, ga)


_bwd_workspaces = _lib.StreamWorkspaces(zero=True)


def boundary_search_backward(energies: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, max_duration: int,
                             log_alpha: torch.Tensor, grad_log_alpha: Optional[torch.Tensor] = None,
                             grad_gamma: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d loss / d energies [B,T_text,T_mel] fp32 for a loss that reads the search's log_alpha and / or gamma:
    `log_alpha` is what boundary_search(..., want_log_alpha=True) returned for the same energies and lengths, the
    two `grad_*` are the loss's cotangents (at least one).  Asynchronous on the current stream."""
    _lib.require_gpu()
    if energies.dim() != 3 or not energies.is_cuda:
        raise ValueError("energies must be a GPU tensor [B, T_text, T_mel]")
    if grad_log_alpha is None and grad_gamma is None:
        raise ValueError("boundary_search_backward needs grad_log_alpha or grad_gamma")
    e = energies.detach()
    if e.dtype not in _DT:
        e = e.float()
    e = e.contiguous()
    B, Tx, Ty = e.shape
    dev = e.device

    def plane(t, name):
        if t is None:
            return None
        if t.shape != (B, Tx, Ty) or t.device != dev:
            raise ValueError(f"{name} must be a [B, T_text, T_mel] tensor on the energies' device")
        return t.detach().to(torch.float32).contiguous()

    la = plane(log_alpha, "log_alpha")
    if la is None:
        raise ValueError("log_alpha (the search's output) is required")
    gla, gga = plane(grad_log_alpha, "grad_log_alpha"), plane(grad_gamma, "grad_gamma")
    tx = torch.as_tensor(t_x).detach().to(device=dev, dtype=torch.int32).contiguous()
    ty = torch.as_tensor(t_y).detach().to(device=dev, dtype=torch.int32).contiguous()
    if tx.shape != (B,) or ty.shape != (B,):
        raise ValueError("t_x / t_y must have one entry per utterance")
    grad = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if B > 0:
        with torch.cuda.device(dev):
            need = lib.aligner_boundary_search_backward_workspace_bytes(B, Tx, Ty, int(max_duration))
            if need == 0:
                need = 256
            ws = _bwd_workspaces.get(dev, need)
            _lib.check(lib.aligner_boundary_search_backward(
                e.data_ptr(), _DT[e.dtype], tx.data_ptr(), ty.data_ptr(), int(max_duration), la.data_ptr(),
                None if gla is None else gla.data_ptr(), None if gga is None else gga.data_ptr(), grad.data_ptr(),
                ws.data_ptr(), ws.numel(), B, Tx, Ty, torch.cuda.current_stream(dev).cuda_stream))
    return grad


class _SoftBoundaries(torch.autograd.Function):
    @staticmethod
    def forward(ctx, energies, t_x, t_y, max_duration, want_map):
        r = boundary_search(energies, t_x, t_y, max_duration, want_log_alpha=True, want_gamma=True, want_map=want_map)
        ctx.save_for_backward(energies, r.log_alpha)
        ctx.lengths = (t_x, t_y, int(max_duration))
        ctx.set_materialize_grads(False)
        if not want_map:
            return r.log_alpha, r.gamma
        ctx.mark_non_differentiable(r.boundaries, r.durations, r.map_score)
        return r.log_alpha, r.gamma, r.boundaries, r.durations, r.map_score

    @staticmethod
    def backward(ctx, g_la, g_ga, *_):
        energies, la = ctx.saved_tensors
        t_x, t_y, D = ctx.lengths
        if g_la is None and g_ga is None:
            return None, None, None, None, None
        grad = boundary_search_backward(energies, t_x, t_y, D, la, g_la, g_ga)
        return grad.to(energies.dtype), None, None, None, None


def soft_boundaries(energies: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, max_duration: int,
                    want_map: bool = True) -> BoundarySearch:
    """boundary_search() with log_alpha and gamma attached to autograd: a loss on either back-propagates into
    `energies` through boundary_search_backward (the MAP outputs carry no gradient; `want_map=False` leaves them out
    and the forward pass runs the sum-product chain alone)."""
    out = _SoftBoundaries.apply(energies, t_x, t_y, max_duration, want_map)
    if not want_map:
        return BoundarySearch(None, None, None, out[0], out[1])
    la, ga, bnd, dur, score = out
    return BoundarySearch(bnd, dur, score, la, ga)


def read_status(device=None) -> int:
    """ALIGNER_ST_* bits left by boundary_search() calls on `device` since the last read (blocking)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = 0
    out = np.zeros(1, np.int32)
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        for ws in list(_workspaces.on_device(device)) + list(_bwd_workspaces.on_device(device)):
            _lib.check(_lib.load().aligner_maxpath_read_status(ws.data_ptr(), out.ctypes.data,
                                                               torch.cuda.current_stream(device).cuda_stream))
            st |= int(out[0])
    return st
