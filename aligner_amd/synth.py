"""Bit-reproducible synthetic alignment inputs.

Scores are dyadic rationals derived from splitmix64, so the same tensor can be
regenerated in numpy, C++ or HIP without libm and hashes identically everywhere.
Used by bench.py, the tests and tests/golden/make_golden.py.  numpy only; no
dependency on the oracle or on the HIP library.
"""
from __future__ import annotations

import hashlib

import numpy as np

_U = np.uint64


def mix(i: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output #i for initial state `seed` (arithmetic mod 2**64)."""
    with np.errstate(over="ignore"):
        z = (i.astype(_U) + _U(1)) * _U(0x9E3779B97F4A7C15) + _U(seed)
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def synth_value(B: int, Tx: int, Ty: int, seed: int, bits: int = 16,
                denom: float = 256.0) -> np.ndarray:
    """value[b,x,y] = -k/denom with k the `bits`-bit field of mix(flat index)."""
    n = B * Tx * Ty
    k = ((mix(np.arange(n, dtype=_U), seed) >> _U(40)) & _U((1 << bits) - 1))
    v = -(k.astype(np.float32) / np.float32(denom))
    return v.reshape(B, Tx, Ty)


def synth_lengths(B: int, Tx_max: int, Ty_min: int, Ty_max: int, seed: int):
    """Variable (t_x, t_y) per utterance: 0.10-0.20 tokens per frame."""
    h = mix(np.arange(B, dtype=_U), seed + 0x1000)
    ty = _U(Ty_min) + h % _U(Ty_max - Ty_min + 1)
    r = ((h >> _U(32)) % _U(11)) + _U(10)
    tx = np.clip(ty * r // _U(100), 1, np.minimum(_U(Tx_max), ty))
    return tx.astype(np.int32), ty.astype(np.int32)


def prefix_mask(tx: np.ndarray, ty: np.ndarray, Tx: int, Ty: int,
                dtype=np.float32) -> np.ndarray:
    """mask[b,x,y] = (x < tx[b]) & (y < ty[b])."""
    xs = np.arange(Tx)[None, :, None] < np.asarray(tx)[:, None, None]
    ys = np.arange(Ty)[None, None, :] < np.asarray(ty)[:, None, None]
    return (xs & ys).astype(dtype)


def sha256_of(a: np.ndarray) -> str:
    """sha256 over the C-contiguous little-endian bytes of `a`."""
    a = np.ascontiguousarray(a)
    if a.dtype.byteorder == ">":
        a = a.byteswap().view(a.dtype.newbyteorder("<"))
    return hashlib.sha256(a.tobytes()).hexdigest()


# The configurations BASELINE.json names, as (B, Tx, Ty, seed, bits, denom).
CONFIGS = {
    "C1": (4, 32, 128, 1, 16, 256.0),
    "C2": (64, 200, 1000, 2, 16, 256.0),
    "C5": (8, 500, 4000, 5, 8, 8.0),
}


def c4_shard(s: int):
    """Shard s (0..7) of the 512-utterance variable-length job (config C4)."""
    B, Tx, Ty = 64, 400, 2000
    value = synth_value(B, Tx, Ty, 40 + s)
    tx_all, ty_all = synth_lengths(512, Tx, 200, Ty, 4)
    sl = slice(64 * s, 64 * s + 64)
    return value, tx_all[sl].copy(), ty_all[sl].copy()


def c4_utterance(g: int):
    """Utterance g (0..511) of the config-C4 job alone: value[400,2000] (the same numbers as c4_shard(g // 64)[0][g % 64])
    and its (t_x, t_y) -- so a rank can materialise exactly the utterances a cost-balanced plan hands it."""
    Tx, Ty = 400, 2000
    s, b = divmod(int(g), 64)
    off = b * Tx * Ty
    k = ((mix(np.arange(off, off + Tx * Ty, dtype=_U), 40 + s) >> _U(40)) & _U(0xFFFF))
    v = -(k.astype(np.float32) / np.float32(256.0))
    tx_all, ty_all = synth_lengths(512, Tx, 200, Ty, 4)
    return v.reshape(Tx, Ty), int(tx_all[g]), int(ty_all[g])
