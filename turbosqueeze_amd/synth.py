"""Deterministic synthetic inputs (libtsq_synth.so, csrc/tsq_synth.c): enwik-shaped text,
random bytes, the 50 % mix and the xorshift32 stream of SURVEY.md 8c.  Host-side numpy arrays."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def _lib():
    global _L
    if _L is None:
        path = os.path.join(HERE, "libtsq_synth.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run __graft_entry__.build()")
        _L = C.CDLL(path)
        _L.tsq_synth_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_double]
        _L.tsq_synth_random.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        _L.tsq_synth_mix.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_double]
        _L.tsq_synth_xorshift32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    return _L


def _buf(n: int, pad: int) -> np.ndarray:
    return np.zeros(n + pad, dtype=np.uint8)


def text(n: int, seed: int = 1, s: float = 0.0, pad: int = 0) -> np.ndarray:
    """enwik-shaped text (oracle no-ext ratio ~0.62); `pad` extra zero bytes are appended."""
    b = _buf(n, pad)
    _lib().tsq_synth_text(b.ctypes.data, n, seed, s)
    return b


def random_bytes(n: int, seed: int = 1, pad: int = 0) -> np.ndarray:
    b = _buf(n, pad)
    _lib().tsq_synth_random(b.ctypes.data, n, seed)
    return b


def mix(n: int, seed: int = 1, pad: int = 0) -> np.ndarray:
    b = _buf(n, pad)
    _lib().tsq_synth_mix(b.ctypes.data, n, seed, 0.0)
    return b


def xorshift32(n: int, seed: int = 2463534242) -> np.ndarray:
    b = _buf(n, 0)
    _lib().tsq_synth_xorshift32(b.ctypes.data, n, seed)
    return b
