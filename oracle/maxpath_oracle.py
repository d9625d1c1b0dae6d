"""oracle/maxpath_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU oracle for the monotonic-alignment hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
aligner_amd/ never does (tests/test_abi.py::test_product_never_imports_the_oracle enforces it).

What is here (reference paths relative to /root/reference):

* maximum_path_c      ctypes binding of oracle/maxpath_oracle.c, the plain-C
                      restatement of monotonic_align/core.pyx:7-45.
* maximum_path        numpy/torch restatement of the Python wrapper
                      monotonic_align/__init__.py:6-21 on top of it.
* column_sweep        an independent numpy restatement (whole-column vector
                      update + 1-bit decisions + bit backtrack) -- the
                      formulation the HIP kernel uses; small sizes only.
* load_ref / load_ref_wrapper
                      the REAL reference: oracle/_ref/core.so is the reference's
                      own core.pyx translated and compiled by oracle/Makefile
                      (`make ref`); load_ref_wrapper additionally executes the
                      reference's __init__.py from /root/reference (authoring
                      container only -- /root/reference does not exist on the
                      GPU box).

Parity status: PINNED against tests/golden/* (made by the real reference via
tests/golden/make_golden.py) and against oracle/_ref when present.
"""
from __future__ import annotations

import ctypes
import importlib.machinery
import importlib.util
import os
import subprocess
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmaxpath_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "core.so")
REFERENCE_ROOT = os.environ.get("ALIGNER_REFERENCE_ROOT", "/root/reference")

_lib = None


def build(ref: bool = True) -> None:
    """Compile the C restatement (and oracle/_ref when the reference tree exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isfile(os.path.join(REFERENCE_ROOT, "monotonic_align", "core.pyx")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref",
                               "REFERENCE=" + REFERENCE_ROOT])


def _load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            build(ref=False)
        lib = ctypes.CDLL(_LIB_PATH)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.oracle_maxpath_c.argtypes = [i32p, f32p, i32p, i32p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_float]
        lib.oracle_maxpath_c.restype = None
        lib.oracle_maxpath_c_omp.argtypes = lib.oracle_maxpath_c.argtypes + [ctypes.c_int]
        lib.oracle_maxpath_c_omp.restype = None
        lib.oracle_has_openmp.restype = ctypes.c_int
        _lib = lib
    return _lib


def _check(paths, values, t_xs, t_ys):
    # Same acceptance rules as the Cython memoryview glue (core.c:27882-27995):
    # exact dtypes, ndim, C-contiguity, writability.
    for name, a, dt, nd in (("paths", paths, np.int32, 3), ("values", values, np.float32, 3),
                            ("t_xs", t_xs, np.int32, 1), ("t_ys", t_ys, np.int32, 1)):
        if not isinstance(a, np.ndarray):
            raise TypeError(f"{name}: expected ndarray")
        if a.ndim != nd:
            raise ValueError(f"Buffer has wrong number of dimensions (expected {nd}, got {a.ndim})")
        if a.dtype != dt:
            raise ValueError(f"Buffer dtype mismatch, expected '{np.dtype(dt).name}' but got '{a.dtype.name}'")
        if not a.flags.c_contiguous:
            raise ValueError("ndarray is not C-contiguous")
        if not a.flags.writeable:
            raise ValueError("buffer source array is read-only")
    if paths.shape != values.shape:
        raise ValueError("paths/values shape mismatch")


def maximum_path_c(paths, values, t_xs, t_ys, max_neg_val=-1e9, num_threads=1):
    """core.pyx:40-45.  Mutates `values` into Q and sets the ones in `paths`."""
    _check(paths, values, t_xs, t_ys)
    lib = _load()
    b, tx, ty = values.shape
    i32p = ctypes.POINTER(ctypes.c_int32)
    f32p = ctypes.POINTER(ctypes.c_float)
    args = (paths.ctypes.data_as(i32p), values.ctypes.data_as(f32p),
            t_xs.ctypes.data_as(i32p), t_ys.ctypes.data_as(i32p),
            b, tx, ty, ctypes.c_float(max_neg_val))
    if num_threads == 1:
        lib.oracle_maxpath_c(*args)
    else:
        lib.oracle_maxpath_c_omp(*args, int(num_threads))


def has_openmp() -> bool:
    return bool(_load().oracle_has_openmp())


def lengths_from_mask(mask: np.ndarray):
    """__init__.py:18-19."""
    t_x = mask.sum(1)[:, 0].astype(np.int32)
    t_y = mask.sum(2)[:, 0].astype(np.int32)
    return t_x, t_y


def maximum_path_numpy(value: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """__init__.py:11-20 on numpy arrays; returns the int32 path [b,t_x,t_y]."""
    value = (value * mask).astype(np.float32)                 # __init__.py:11,14
    value = np.ascontiguousarray(value)
    path = np.zeros_like(value).astype(np.int32)              # __init__.py:15
    t_x, t_y = lengths_from_mask(np.asarray(mask))            # __init__.py:18-19
    maximum_path_c(path, value, np.ascontiguousarray(t_x), np.ascontiguousarray(t_y))
    return path


def maximum_path(value, mask):
    """Torch-level restatement of __init__.py:6-21 (same dtype/device rules)."""
    import torch
    value = value * mask                                      # :11
    device, dtype = value.device, value.dtype                 # :12-13
    v = value.data.cpu().numpy().astype(np.float32)           # :14
    path = np.zeros_like(v).astype(np.int32)                  # :15
    m = mask.data.cpu().numpy()                               # :16
    t_x = m.sum(1)[:, 0].astype(np.int32)                     # :18
    t_y = m.sum(2)[:, 0].astype(np.int32)                     # :19
    maximum_path_c(path, v, t_x, t_y)                         # :20
    return torch.from_numpy(path).to(device=device, dtype=dtype)  # :21


def column_sweep(value: np.ndarray, t_x: int, t_y: int, max_neg_val: float = -1e9):
    """Independent restatement in the kernel's formulation.

    One vector update per mel frame over all text rows (no band test: cells
    outside the band are computed but provably never read by in-band cells),
    a 1-bit decision per cell, and a backtrack over the bits only.  Returns
    (tok[t_y] int32 token index per frame, dec[t_x,t_y] bool).  Pure numpy:
    keep sizes small.
    """
    value = np.asarray(value, dtype=np.float32)
    neg = np.float32(max_neg_val)
    xs = np.arange(t_x)
    q = np.full(t_x, neg, dtype=np.float32)       # Q[:, -1]; content irrelevant
    dec = np.zeros((t_x, t_y), dtype=bool)
    with np.errstate(all="ignore"):
        for y in range(t_y):
            up = np.empty(t_x, dtype=np.float32)
            up[1:] = q[:-1]
            up[0] = np.float32(0.0) if y == 0 else neg       # core.pyx:23-27
            cur = np.where(xs == y, neg, q)                  # core.pyx:19-22
            adv = up > cur                                   # core.c:19384
            q = (np.where(adv, up, cur) + value[:t_x, y]).astype(np.float32)
            dec[:, y] = (xs != 0) & ((xs == y) | adv)        # core.pyx:34 predicate
    tok = np.zeros(t_y, dtype=np.int32)
    idx = t_x - 1
    for y in range(t_y - 1, -1, -1):
        tok[y] = idx
        if dec[idx, y]:
            idx -= 1
    return tok, dec


def path_from_tok(tok: np.ndarray, t_x_pad: int, t_y_pad: int) -> np.ndarray:
    p = np.zeros((t_x_pad, t_y_pad), dtype=np.int32)
    p[tok, np.arange(len(tok))] = 1
    return p


# --------------------------------------------------------------------------- #
# The real reference (authoring container only: oracle/_ref is git- AND gpurun-ignored)
# --------------------------------------------------------------------------- #
def load_ref():
    """Return the compiled reference extension module (oracle/_ref/core.so) or None."""
    if not os.path.isfile(_REF_SO):
        return None
    name = "_aligner_ref_pkg.monotonic_align.core"
    if name in sys.modules:
        return sys.modules[name]
    loader = importlib.machinery.ExtensionFileLoader(name, _REF_SO)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules[name] = mod
    return mod


def load_ref_wrapper():
    """Execute the reference's own __init__.py (where it lies) on top of
    oracle/_ref/core.so and return its `maximum_path`; None if unavailable."""
    core = load_ref()
    init_py = os.path.join(REFERENCE_ROOT, "monotonic_align", "__init__.py")
    if core is None or not os.path.isfile(init_py):
        return None
    pkg_name = "_aligner_ref_pkg"
    if pkg_name in sys.modules and hasattr(sys.modules[pkg_name], "maximum_path"):
        return sys.modules[pkg_name].maximum_path
    sub = types.ModuleType(pkg_name + ".monotonic_align")
    sub.__path__ = []                      # namespace stand-in holding only `core`
    sub.core = core
    sys.modules[pkg_name + ".monotonic_align"] = sub
    spec = importlib.util.spec_from_file_location(
        pkg_name, init_py, submodule_search_locations=[])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[pkg_name] = pkg
    spec.loader.exec_module(pkg)
    return pkg.maximum_path
