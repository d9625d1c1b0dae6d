"""Host-side mirror of the reference's monotonic_align interface on MI355X.

    maximum_path(value, mask)                 reference monotonic_align/__init__.py:6-21
    maximum_path_c(paths, values, t_xs, t_ys) reference monotonic_align/core.pyx:38-45
    align(value, t_x, t_y, ...)               the native entry point: lengths instead
                                              of a dense mask, durations / per-frame
                                              token index without the dense path

Everything is computed by the HIP kernels behind include/aligner_amd.h; PyTorch is
only used for device memory and streams.  There is no CPU implementation here:
CPU tensors are staged through the GPU and the call raises if no GPU is visible.
"""
from __future__ import annotations

import os
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib

_TORCH_TO_DT = {
    torch.float32: _lib.DT_F32, torch.float16: _lib.DT_F16, torch.bfloat16: _lib.DT_BF16,
    torch.float64: _lib.DT_F64, torch.int32: _lib.DT_I32, torch.uint8: _lib.DT_U8,
    torch.bool: _lib.DT_U8, torch.int64: _lib.DT_I64,
}

# score dtypes the kernels read directly (16-bit ones are up-cast inside the loaders: no conversion pass)
_SCORE_DTYPES = (torch.float32, torch.bfloat16, torch.float16)

_workspaces = _lib.StreamWorkspaces(zero=True)     # status word: zero at allocation, sticky afterwards
_CHECK_DEFAULT = os.environ.get("ALIGNER_AMD_CHECK", "") not in ("", "0")


class Alignment(NamedTuple):
    path: Optional[torch.Tensor]        # [B,Tx,Ty] 0/1 in the requested dtype
    tok: Optional[torch.Tensor]         # [B,Ty] int32, token index per frame, -1 past t_y
    durations: Optional[torch.Tensor]   # [B,Tx] int32 frames per token


def _device_for(t: torch.Tensor) -> torch.device:
    _lib.require_gpu()
    if t.is_cuda:
        return t.device
    return torch.device("cuda", torch.cuda.current_device())


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    return _workspaces.get(device, nbytes)             # one per (device, stream): calls on two streams never share


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def align(value: torch.Tensor, t_x: Optional[torch.Tensor] = None, t_y: Optional[torch.Tensor] = None, *,
          mask: Optional[torch.Tensor] = None, strict_mask: bool = False,
          want_path: bool = True, path_dtype: Optional[torch.dtype] = None,
          want_tok: bool = False, want_durations: bool = True,
          max_neg_val: float = -1e9, compat_tx_gt_ty: bool = False,
          force_generic: bool = False, no_prev_table: bool = False, stream_path: bool = False,
          cus_per_utterance: Optional[int] = None,
          out_path: Optional[torch.Tensor] = None, out_path_is_zero: bool = False,
          check: Optional[bool] = None, _test_flags: int = 0) -> Alignment:
    """Monotonic alignment search for a batch resident on the GPU.

    value [B,Tx,Ty] float (computed in fp32 like the reference, __init__.py:14; fp32, bf16 and fp16 tensors are
    read as they are, other dtypes are cast to fp32 first);
    lengths either as int32 vectors t_x/t_y [B] or derived from `mask`
    (__init__.py:18-19).  With strict_mask the scores are first multiplied by the
    mask element-wise (__init__.py:11).  Asynchronous on the current stream.
    stream_path: write the dense path with non-temporal stores (ALIGNER_F_STREAM_PATH: worth it when other batches
    are in flight on the GPU and nothing reads the path back at once).
    cus_per_utterance: None = the library decides (long text on a long mel axis in a small batch runs as two
    workgroups per utterance, ALIGNER_F_ONE_CU / ALIGNER_F_TWO_CUS otherwise); the results do not depend on it.
    out_path / out_path_is_zero: write the path into the caller's tensor; with out_path_is_zero the caller vouches
    that it already holds zeros (e.g. written on another stream while the scores were computed): the search kernel
    then writes only the ones (ALIGNER_F_PATH_PREZEROED) and the 4*B*Tx*Ty-byte expand pass is not launched.
    check: wait for the call and raise RuntimeError if the device's status word reports ALIGNER_ST_INTERNAL
    (an internal consistency check failed: e.g. the two-workgroup form gave up waiting for its other half --
    the outputs of that utterance are then all-zero, never a wrong path).  Default: the environment variable
    ALIGNER_AMD_CHECK=1, else off -- the call stays asynchronous, and read_status() reports the bit later.
    """
    if cus_per_utterance not in (None, 1, 2):
        raise ValueError("cus_per_utterance must be None, 1 or 2")
    if value.dim() != 3:
        raise ValueError(f"value must be [b, t_x, t_y], got {tuple(value.shape)}")
    if not value.is_cuda:
        raise ValueError("align() takes GPU tensors; use maximum_path() for CPU tensors")
    device = value.device
    B, Tx, Ty = value.shape
    lib = _lib.load()
    with torch.no_grad(), torch.cuda.device(device):
        v = value.detach()
        m = None
        mdt = 0
        if mask is not None:
            if mask.shape != value.shape:
                raise ValueError("mask and value must have the same shape")
            m = mask.detach()
        if strict_mask and m is not None and (m.dtype != v.dtype or v.dtype not in _SCORE_DTYPES):
            # torch's type promotion decides the dtype the product is rounded in: take the product exactly as
            # the reference does (__init__.py:11); the mask then only supplies the lengths
            v = v * m
            strict_mask = False
        if v.dtype not in _SCORE_DTYPES:
            v = v.float()                   # fp64 / integer scores: the reference's astype(np.float32)
        # a row pitch of its own (softattn.pitched_logp(): rows on whole 128-byte lines) is read as it is -- with lengths,
        # without a mask; anything else that is not contiguous is copied
        ld = Ty
        ldc = int(v.stride(1)) if Tx > 1 else int(v.stride(0))
        if (not v.is_contiguous() and m is None and t_x is not None and t_y is not None and v.stride(2) == 1 and ldc >= Ty and
                (B == 1 or v.stride(0) == Tx * ldc) and (ldc * v.element_size()) % 16 == 0):
            ld = ldc
        elif not v.is_contiguous():
            v = v.contiguous()
        if m is not None:
            if m.dtype not in (torch.float32, torch.bfloat16, torch.float16, torch.uint8, torch.bool, torch.int32):
                m = m.float()
            if not m.is_contiguous():
                m = m.contiguous()
            mdt = _TORCH_TO_DT[m.dtype]
        if (t_x is None or t_y is None) and m is None:
            raise ValueError("need t_x/t_y or a mask")
        if strict_mask and m is None:
            raise ValueError("strict_mask needs a mask")
        if t_x is not None and t_y is not None:
            t_x = t_x.to(device=device, dtype=torch.int32).contiguous()
            t_y = t_y.to(device=device, dtype=torch.int32).contiguous()
            if t_x.numel() != B or t_y.numel() != B:
                raise ValueError("t_x/t_y must have one entry per utterance")
        else:
            t_x = t_y = None
        path = None
        pdt = 0
        if want_path:
            pd = path_dtype or value.dtype
            if out_path is not None:
                path = out_path
                if path.shape != value.shape or not path.is_contiguous() or path.dtype not in _TORCH_TO_DT:
                    raise ValueError("out_path must be contiguous, same shape, supported dtype")
                pdt = _TORCH_TO_DT[path.dtype]
            elif pd in _TORCH_TO_DT:
                path = torch.empty((B, Tx, Ty), dtype=pd, device=device)
                pdt = _TORCH_TO_DT[pd]
            else:
                path = torch.empty((B, Tx, Ty), dtype=torch.float32, device=device)
                pdt = _lib.DT_F32
        tok = torch.empty((B, Ty), dtype=torch.int32, device=device) if want_tok else None
        dur = torch.empty((B, Tx), dtype=torch.int32, device=device) if want_durations else None
        if B > 0 and Tx > 0 and Ty > 0:
            nbytes = lib.aligner_maxpath_workspace_bytes(B, Tx, Ty)
            ws = _workspace(device, nbytes)
            flags = (_lib.F_STRICT_MASK if strict_mask else 0) | \
                    (_lib.F_COMPAT_TXGTTY if compat_tx_gt_ty else 0) | \
                    (_lib.F_FORCE_GENERIC if force_generic else 0) | \
                    (_lib.F_NO_PREV_TABLE if no_prev_table else 0) | \
                    (_lib.F_STREAM_PATH if stream_path else 0) | \
                    (_lib.F_ONE_CU if cus_per_utterance == 1 else _lib.F_TWO_CUS if cus_per_utterance == 2 else 0) | \
                    (_lib.F_PATH_PREZEROED if (out_path_is_zero and out_path is not None) else 0) | \
                    int(_test_flags)
            if ld != Ty:
                _lib.check(lib.aligner_maxpath_ld(
                    v.data_ptr(), _TORCH_TO_DT[v.dtype], ld, _ptr(t_x), _ptr(t_y), _ptr(path), pdt, _ptr(tok), _ptr(dur),
                    ws.data_ptr(), ws.numel(), B, Tx, Ty, float(max_neg_val), flags, _stream_ptr(device)))
            else:
                _lib.check(lib.aligner_maxpath(
                    v.data_ptr(), _TORCH_TO_DT[v.dtype], _ptr(m), mdt, _ptr(t_x), _ptr(t_y), _ptr(path), pdt, _ptr(tok), _ptr(dur),
                    ws.data_ptr(), ws.numel(), B, Tx, Ty, float(max_neg_val), flags, _stream_ptr(device)))
        else:
            for t in (path, tok, dur):
                if t is not None:
                    t.zero_()
        if want_path and out_path is None and path.dtype != (path_dtype or value.dtype):
            path = path.to(path_dtype or value.dtype)
        if check is None:
            check = _CHECK_DEFAULT
        if check and B > 0 and Tx > 0 and Ty > 0:
            st = _status_of(ws, device)
            if st & _lib.ST_INTERNAL:
                raise RuntimeError(f"aligner_amd: internal consistency check failed on {device} (status word {st}); "
                                   "the affected utterances were returned all-zero")
    return Alignment(path, tok, dur)


def _status_of(ws: torch.Tensor, device: torch.device) -> int:
    """Blocking read of one workspace's status word WITHOUT clearing it (read_status() still sees the bits)."""
    torch.cuda.current_stream(device).synchronize()
    return int(ws[:4].view(torch.int32).item())


def release_workspaces(device=None, stream=None) -> None:
    """Free the scratch buffers align()/maximum_path() keep per (device, stream); see _lib.StreamWorkspaces."""
    _workspaces.release(device, stream)


def read_status(device=None) -> int:
    """ALIGNER_ST_* bits left by align()/maximum_path() calls on `device` since the last read, over all
    streams (blocking: synchronises the device first)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = 0
    out = np.zeros(1, np.int32)
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        for ws in _workspaces.on_device(device):
            _lib.check(_lib.load().aligner_maxpath_read_status(ws.data_ptr(), out.ctypes.data, _stream_ptr(device)))
            st |= int(out[0])
    return st


_ws_bytes: dict = {}            # (B, Tx, Ty) -> aligner_maxpath_workspace_bytes


def _maximum_path_resident(value: torch.Tensor, mask: torch.Tensor, mask_is_prefix: bool) -> torch.Tensor:
    """maximum_path() for what a training step passes: two contiguous GPU tensors of one floating dtype the kernels
    read as it is.  One C-ABI call (the mask is verified / multiplied on the device, the lengths come from it, the dense
    path is written by the search launch) and nothing else on the host: no promotion logic, no copies, no per-call
    driver queries."""
    device = value.device
    B, Tx, Ty = value.shape
    path = torch.empty((B, Tx, Ty), dtype=value.dtype, device=device)
    if B == 0 or Tx == 0 or Ty == 0:
        return path
    lib = _lib.load()
    other = torch.cuda.current_device() != device.index
    if other:
        prev = torch.cuda.current_device()
        torch.cuda.set_device(device)
    try:
        nbytes = _ws_bytes.get((B, Tx, Ty))
        if nbytes is None:
            nbytes = _ws_bytes[(B, Tx, Ty)] = lib.aligner_maxpath_workspace_bytes(B, Tx, Ty)
        ws = _workspaces.get(device, nbytes)
        dt = _TORCH_TO_DT[value.dtype]
        rc = lib.aligner_maxpath(value.data_ptr(), dt, mask.data_ptr(), dt, None, None, path.data_ptr(), dt, None, None,
                                 ws.data_ptr(), ws.numel(), B, Tx, Ty, -1e9,
                                 (0 if mask_is_prefix else _lib.F_STRICT_MASK) | _lib.F_COMPAT_TXGTTY,
                                 torch.cuda.current_stream(device).cuda_stream)
        if rc:
            _lib.check(rc)
    finally:
        if other:
            torch.cuda.set_device(prev)
    if _CHECK_DEFAULT:
        st = _status_of(ws, device)
        if st & _lib.ST_INTERNAL:
            raise RuntimeError(f"aligner_amd: internal consistency check failed on {device} (status word {st}); "
                               "the affected utterances were returned all-zero")
    return path


def maximum_path(value: torch.Tensor, mask: torch.Tensor, *, mask_is_prefix: bool = False) -> torch.Tensor:
    """Drop-in for the reference's maximum_path (monotonic_align/__init__.py:6-21).

    value: [b, t_x, t_y], mask: [b, t_x, t_y].  Returns the 0/1 path with the
    dtype of `value * mask` on value's device, requires_grad False -- the same
    contract as the reference.  Differences, all supersets: bf16 works (the
    reference raises TypeError because numpy has no bf16) and nothing round-trips
    through host memory for GPU tensors.

    mask_is_prefix=True promises a 0/1 prefix-rectangle mask (the usual
    x_mask[:, :, None] * y_mask[:, None, :]) and skips the element-wise multiply:
    the DP provably never reads a masked cell, so the result is identical while
    the mask is only read for the lengths.
    """
    if value.dim() != 3 or mask.dim() != 3:
        raise ValueError("Buffer has wrong number of dimensions (expected 3)")   # core.c:27882
    if (value.is_cuda and value.dtype is mask.dtype and value.dtype in _SCORE_DTYPES and value.shape == mask.shape
            and value.device == mask.device and value.is_contiguous() and mask.is_contiguous()):
        return _maximum_path_resident(value, mask, mask_is_prefix)
    out_dtype = torch.result_type(value, mask)                                   # value * mask, :11
    if not out_dtype.is_floating_point:
        # the reference would multiply in an integer dtype and then cast to fp32
        out_dtype_run = torch.float32
    else:
        out_dtype_run = out_dtype
    shape = torch.broadcast_shapes(value.shape, mask.shape)
    if tuple(shape) != tuple(value.shape) or tuple(mask.shape) != tuple(value.shape):
        value, mask = torch.broadcast_tensors(value, mask)
    if not (value.is_contiguous() and mask.is_contiguous()):
        # value*mask keeps non-C strides and the memoryview rejects them (core.c:27843)
        raise ValueError("ndarray is not C-contiguous")
    src_device = value.device
    device = _device_for(value)
    v = value.detach().to(device)
    m = mask.detach().to(device)
    res = align(v, mask=m, strict_mask=not mask_is_prefix, want_path=True, path_dtype=out_dtype_run,
                want_tok=False, want_durations=False,
                compat_tx_gt_ty=True)
    path = res.path
    if path.dtype != out_dtype:
        path = path.to(out_dtype)
    if src_device != device:
        path = path.to(src_device)
    return path


def maximum_path_c(paths: np.ndarray, values: np.ndarray, t_xs: np.ndarray, t_ys: np.ndarray,
                   max_neg_val: float = -1e9, write_q: bool = True) -> None:
    """Drop-in for monotonic_align.core.maximum_path_c (core.pyx:40-45) on numpy buffers.

    Same argument checks as the Cython memoryview glue (exact dtypes, ndim,
    C-contiguity, writability -> ValueError).  `paths` is overwritten with the
    path and, as in the reference, `values` with the running score Q inside every
    utterance's band (core.pyx:30), bit for bit.  write_q=False leaves `values`
    untouched and takes the fast kernel (nothing downstream of the reference's own
    wrapper reads Q, __init__.py:20-21).
    """
    for a, dt, nd in ((paths, np.int32, 3), (values, np.float32, 3), (t_xs, np.int32, 1), (t_ys, np.int32, 1)):
        if not isinstance(a, np.ndarray):
            raise TypeError("expected a numpy array")
        if a.ndim != nd:
            raise ValueError(f"Buffer has wrong number of dimensions (expected {nd}, got {a.ndim})")
        if a.dtype != dt:
            raise ValueError(f"Buffer dtype mismatch, expected '{np.dtype(dt).name}' but got '{a.dtype.name}'")
        if not a.flags.c_contiguous:
            raise ValueError("ndarray is not C-contiguous")
        if not a.flags.writeable:
            raise ValueError("buffer source array is read-only")
    if paths.shape != values.shape or t_xs.shape[0] != values.shape[0] or t_ys.shape[0] != values.shape[0]:
        raise ValueError("shape mismatch between paths/values/t_xs/t_ys")
    _lib.require_gpu()
    b, tx, ty = values.shape
    if b == 0 or tx == 0 or ty == 0:
        return
    rc = _lib.load().aligner_maxpath_host_f32(paths.ctypes.data, values.ctypes.data, t_xs.ctypes.data,
                                               t_ys.ctypes.data, b, tx, ty, float(max_neg_val),
                                               _lib.F_WRITE_Q if write_q else 0)
    if rc == -33:
        raise ValueError(_lib.load().aligner_last_error().decode())
    _lib.check(rc)
