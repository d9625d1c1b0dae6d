"""Experiments kept for their measurements -- NOT part of the default API surface.

`fused_align` is SURVEY.md 8f rank 1 (similarity -> log-softmax -> alignment search in ONE kernel, the log-probabilities
consumed from LDS as the matrix cores produce them).  It is built, bit-exact against `align()` on its own log-probs
(tests/test_fused.py) -- and CLOSED as "measured, loses" (DESIGN_HISTORY.md 7.2): 126 us against 53 us for the two
kernels at [64,80,200,1000], 52 against 36 us per step with four batches in flight, because all of `logp` then leaves
through the 64 CUs the search runs on (800 KB per CU at the 7-11 B/clk a CU stores: 30-47 us by itself).  It is limited to
C <= 80, T_text <= 252, T_mel <= 2048, no prior.  Use `soft_attention()` + `align()`; import this only to re-measure.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .maxpath import Alignment, _TORCH_TO_DT, _ptr, _stream_ptr, _workspace


def fused_align(keys_enc: torch.Tensor, queries_enc: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor, *,
                temperature: float = 0.0005, sim: str = "l2", want_logp: bool = True, want_path: bool = True,
                path_dtype: Optional[torch.dtype] = None, want_tok: bool = False, max_neg_val: float = -1e9):
    """soft_attention() + align() in one kernel: the DP consumes the log-probabilities from
    LDS as they are produced; logp is written once (or not at all) and never re-read.  Returns (logp or None,
    Alignment).  The path equals align() run on the returned logp, bit for bit.
    Limits of this form: C <= 80, T_text <= 252, T_mel <= 2048, finite encodings, no prior."""
    _lib.require_gpu()
    k = keys_enc.detach().float().contiguous()
    q = queries_enc.detach().float().contiguous()
    if not (k.is_cuda and q.is_cuda):
        raise ValueError("fused_align() takes GPU tensors")
    B, C, Tx = k.shape
    B2, C2, Ty = q.shape
    if B != B2 or C != C2:
        raise ValueError("keys/queries shape mismatch")
    dev = k.device
    tx = t_x.to(device=dev, dtype=torch.int32).contiguous()
    ty = t_y.to(device=dev, dtype=torch.int32).contiguous()
    logp = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev) if want_logp else None
    tok = torch.empty((B, Ty), dtype=torch.int32, device=dev) if want_tok else None
    dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
    pd = path_dtype or torch.float32
    path = torch.empty((B, Tx, Ty), dtype=pd, device=dev) if want_path else None
    lib = _lib.load()
    simc = {"l2": _lib.SIM_L2, "dot": _lib.SIM_DOT}[sim]
    if B > 0:
        with torch.cuda.device(dev):
            ws = _workspace(dev, lib.aligner_maxpath_workspace_bytes(B, Tx, Ty))
            s = _stream_ptr(dev)
            _lib.check(lib.aligner_fused_align_f32(k.data_ptr(), q.data_ptr(), tx.data_ptr(), ty.data_ptr(), _ptr(logp),
                                                   _ptr(tok), dur.data_ptr(), ws.data_ptr(), ws.numel(), B, C, Tx, Ty,
                                                   float(temperature), simc, float(max_neg_val), s))
            if path is not None:
                _lib.check(lib.aligner_maxpath_expand(ws.data_ptr(), path.data_ptr(), _TORCH_TO_DT[pd], B, Tx, Ty, s))
    return logp, Alignment(path, tok, dur)
