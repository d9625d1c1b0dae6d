"""ctypes binding of libaligner_amd.so (the C ABI declared in include/aligner_amd.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C aligner_amd/csrc`
and is the ONLY implementation of the hot path: there is no CPU or PyTorch
fallback anywhere in this package.  If the library is missing, or there is no
GPU, the product entry points raise.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALIGNER_AMD_LIB: development override (A/B builds of the same library); the default is the in-tree build
LIB_PATH = os.environ.get("ALIGNER_AMD_LIB") or os.path.join(_HERE, "lib", "libaligner_amd.so")

# dtype codes (include/aligner_amd.h)
DT_F32, DT_F16, DT_BF16, DT_F64, DT_I32, DT_U8, DT_I64 = range(7)

F_STRICT_MASK = 1
F_COMPAT_TXGTTY = 2
F_FORCE_GENERIC = 4
F_NO_PREV_TABLE = 16
F_WRITE_Q = 64
F_STREAM_PATH = 128
F_ONE_CU = 256
F_TWO_CUS = 512
F_TEST_DROP_FIRST_HALF = 1024
F_TEST_IMPATIENT_FIRST_HALF = 16384
F_PATH_PREZEROED = 2048
F_SEPARATE_EXPAND = 4096
F_TEST_DROP_ZERO_REPORTS = 8192

ST_BAD_LENGTHS = 1
ST_CLAMPED = 2
ST_INTERNAL = 4

SIM_L2 = 0
SIM_DOT = 1

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t


class ConvLayer(ctypes.Structure):
    """struct aligner_conv_layer (include/aligner_amd.h): one layer of an encoder stack."""
    _fields_ = [("prepared", _c.c_void_p), ("bias", _c.c_void_p),
                ("Cin", _c.c_int), ("Cout", _c.c_int), ("K", _c.c_int), ("relu", _c.c_int)]


# name -> (restype, argtypes); every symbol include/aligner_amd.h declares
SIGNATURES = {
    "aligner_abi_version": (_i, []),
    "aligner_last_error": (_c.c_char_p, []),
    "aligner_device_count": (_i, []),
    "aligner_maxpath_workspace_bytes": (_sz, [_i, _i, _i]),
    "aligner_lengths_from_mask": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "aligner_maxpath_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz,
                                 _i, _i, _i, _f, _i, _vp]),
    "aligner_maxpath": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz,
                             _i, _i, _i, _f, _i, _vp]),
    "aligner_maxpath_ld": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _i, _vp]),
    "aligner_maxpath_forward": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                     _i, _i, _i, _f, _i, _vp]),
    "aligner_maxpath_forward_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                         _i, _i, _i, _f, _i, _vp]),
    "aligner_maxpath_expand": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "aligner_maxpath_expand_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aligner_maxpath_zero_path": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "aligner_maxpath_scatter_path": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "aligner_maxpath_read_status": (_i, [_vp, _vp, _vp]),
    "aligner_debug_set_stamps": (None, [_vp]),
    "aligner_debug_set_option": (_i, [_c.c_char_p, _i]),
    "aligner_maxpath_host_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "aligner_softattn_workspace_bytes": (_sz, [_i, _i, _i]),
    "aligner_softattn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _i, _f, _i, _vp]),
    "aligner_softattn_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _f, _i, _vp]),
    "aligner_softattn_ld": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _i, _i, _i, _i, _f, _i, _vp]),
    "aligner_conv1d_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aligner_conv1d_prepared_bytes": (_sz, [_i, _i, _i]),
    "aligner_conv1d_prepare_f32": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp]),
    "aligner_conv1d_prepared_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aligner_conv1d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "aligner_conv1d_prepared_ws_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "aligner_conv_stack_workspace_bytes": (_sz, [_c.POINTER(ConvLayer), _i, _i, _i]),
    "aligner_conv_stack_f32": (_i, [_vp, _c.POINTER(ConvLayer), _i, _vp, _vp, _sz, _i, _i, _vp]),
    "aligner_forward_sum_workspace_bytes": (_sz, [_i, _i, _i]),
    "aligner_forward_sum_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "aligner_forward_sum_ctc_workspace_bytes": (_sz, [_i, _i, _i]),
    "aligner_forward_sum_ctc_f32": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "aligner_beta_binomial_prior_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "aligner_boundary_search_workspace_bytes": (_sz, [_i, _i, _i]),
    "aligner_boundary_search_workspace_bytes_ex": (_sz, [_i, _i, _i, _i]),
    "aligner_boundary_search": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "aligner_boundary_search_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "aligner_boundary_search_backward": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "aligner_regulate_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}


EINVAL, EDOM, ENOSPC = -22, -33, -28     # ALIGNER_E* (include/aligner_amd.h)


class AlignerError(RuntimeError):
    """A C-ABI call returned a negative ALIGNER_E* code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libaligner_amd error {code}: {message}")
        self.code = code


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library, failing loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C aligner_amd/csrc`. aligner_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)            # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        # development: ALIGNER_DEBUG_OPTIONS="name=value,..." sets the library's A-B / testing switches
        # (aligner_debug_set_option) for a whole process, e.g. under rocprofv3
        for item in filter(None, os.environ.get("ALIGNER_DEBUG_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if lib.aligner_debug_set_option(name.strip().encode(), int(value or "1")) != 0:
                raise ValueError(f"ALIGNER_DEBUG_OPTIONS: unknown option {name!r}")
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().aligner_last_error()
        raise AlignerError(rc, msg.decode("utf-8", "replace") if msg else "")


def require_gpu() -> None:
    lib = load()
    if lib.aligner_device_count() < 1:
        raise RuntimeError("aligner_amd needs an AMD GPU (gfx950): no HIP device is visible and "
                           "there is deliberately no CPU fallback")


class StreamWorkspaces:
    """Scratch buffers for the C-ABI calls, one per (device, stream).

    Every entry point is asynchronous on the caller's current stream and its workspace holds
    live state between the launches of one call (token starts, decision words, status word), so
    two calls in flight on different streams of one device must never share a buffer.  Buffers
    come from torch's caching allocator while their stream is current.

    Growing: a larger buffer replaces the old one for later calls; the old one's sticky status word is
    OR-ed into the new one on the stream (read_status() promises "since the last read"), and the old
    buffer is kept alive in `retired` until release() -- a HIP graph captured through align() still
    launches into it.  Entries are keyed by the raw stream handle and live until release(): callers that
    capture graphs or create streams at a high rate should own their workspaces through the C ABI
    (bench.py does) or call release(stream) when a stream is done."""

    def __init__(self, zero: bool, slack: float = 1.25):
        self.zero = zero
        self.slack = slack
        self.bufs: dict = {}
        self.retired: dict = {}

    def get(self, device, nbytes: int):
        import torch
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = self.bufs.get(key)
        if ws is None or ws.numel() < nbytes:
            n = int(nbytes * self.slack) + 256
            with torch.cuda.device(device):
                new = (torch.zeros if self.zero else torch.empty)(n, dtype=torch.uint8, device=device)
                if ws is not None:
                    if self.zero:                     # carry the sticky status word over (stream-ordered)
                        new[:4].view(torch.int32).bitwise_or_(ws[:4].view(torch.int32))
                    self.retired.setdefault(key, []).append(ws)
            self.bufs[key] = ws = new
        return ws

    def on_device(self, device):
        return [ws for (d, _), ws in self.bufs.items() if d == device]

    def release(self, device=None, stream=None) -> None:
        """Drop the buffers of one stream (or of every stream of `device`, or all).  The caller vouches
        that no work -- and no captured graph -- still uses them."""
        for key in list(self.bufs):
            d, s = key
            if (device is None or d == device) and (stream is None or s == stream):
                self.bufs.pop(key, None)
                self.retired.pop(key, None)
