"""ctypes binding of the CPU oracle (oracle/libtsq_oracle.so) and, when present,
of the compiled reference (oracle/_ref/libtsq_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by turbosqueeze_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK_SZ = 1 << 22
OUTPUT_SZ = BLOCK_SZ + (BLOCK_SZ >> 2)
HASH_ENTRIES = 1 << 17
HALO = 128

_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> None:
    """Compile the checker (gcc) and, if /root/reference is here, oracle/_ref."""
    so = os.path.join(HERE, "libtsq_oracle.so")
    src = os.path.join(HERE, "tsq_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "libtsq_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(HERE, "_ref", "libtsq_ref.so")
    if os.path.exists("/root/reference/tsq_encode.cpp") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


class Oracle:
    """Our C restatement (tsq_oracle.c)."""

    def __init__(self) -> None:
        build()
        L = C.CDLL(os.path.join(HERE, "libtsq_oracle.so"))
        L.tsqo_bound.restype = C.c_uint32
        L.tsqo_bound.argtypes = [C.c_uint32]
        L.tsqo_encode_block.restype = C.c_uint32
        L.tsqo_encode_block.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_void_p]
        L.tsqo_decode_block.restype = C.c_uint32
        L.tsqo_decode_block.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_int)]
        L.tsqo_compress_bound.restype = C.c_size_t
        L.tsqo_compress_bound.argtypes = [C.c_size_t]
        L.tsqo_compress.restype = C.c_size_t
        L.tsqo_compress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_uint32, C.c_int]
        L.tsqo_decompressed_size.restype = C.c_size_t
        L.tsqo_decompressed_size.argtypes = [_u8p, C.c_size_t]
        L.tsqo_decompress.restype = C.c_size_t
        L.tsqo_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int]
        L.tsqo_fnv1a64.restype = C.c_uint64
        L.tsqo_fnv1a64.argtypes = [_u8p, C.c_size_t]
        self.L = L
        self._table = np.zeros(HASH_ENTRIES, dtype=np.uint16)

    @staticmethod
    def _with_halo(data, halo=None) -> np.ndarray:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        buf = np.zeros(a.size + HALO, dtype=np.uint8)
        buf[: a.size] = a
        if halo is not None:
            h = np.frombuffer(bytes(halo), dtype=np.uint8)[:HALO]
            buf[a.size : a.size + h.size] = h
        return buf

    def encode_block(self, data, ext: int, halo=None) -> bytes:
        """One block, canonical conditions; `halo` = the bytes that follow it (zeros if None)."""
        n = len(data)
        assert n <= BLOCK_SZ
        buf = self._with_halo(data, halo)
        out = np.empty(self.L.tsqo_bound(n) + 32, dtype=np.uint8)
        sz = self.L.tsqo_encode_block(_ptr(buf), n, _ptr(out), int(ext), self._table.ctypes.data)
        return out[:sz].tobytes()

    def decode_block(self, stream, ext: int):
        s = np.frombuffer(bytes(stream), dtype=np.uint8)
        out = np.zeros(BLOCK_SZ + 16, dtype=np.uint8)
        st = C.c_int(0)
        n = self.L.tsqo_decode_block(_ptr(s), s.size, _ptr(out), int(ext), C.byref(st))
        return out[:n].tobytes(), st.value

    def compress(self, data, ext: int, threads: int = 1) -> bytes:
        """Whole .tsq container (header + frames), blocks contiguous (block k's halo = block k+1)."""
        a = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
        buf = self._with_halo(a)
        out = np.empty(self.L.tsqo_compress_bound(a.size), dtype=np.uint8)
        sz = self.L.tsqo_compress(_ptr(buf), a.size, _ptr(out), int(ext), int(threads))
        return out[:sz].tobytes()

    def compress_into(self, buf_with_halo: np.ndarray, n: int, out: np.ndarray, ext: int, threads: int) -> int:
        return self.L.tsqo_compress(_ptr(buf_with_halo), n, _ptr(out), int(ext), int(threads))

    def decompress(self, blob, threads: int = 1):
        s = blob if isinstance(blob, np.ndarray) else np.frombuffer(bytes(blob), dtype=np.uint8)
        total = self.L.tsqo_decompressed_size(_ptr(s), s.size)
        if total == C.c_size_t(-1).value:
            return None
        out = np.empty(total + 16, dtype=np.uint8)
        got = self.L.tsqo_decompress(_ptr(s), s.size, _ptr(out), total, int(threads))
        if got == C.c_size_t(-1).value:
            return None
        return out[:got].tobytes()

    def decompress_into(self, s: np.ndarray, out: np.ndarray, threads: int) -> int:
        return self.L.tsqo_decompress(_ptr(s), s.size, _ptr(out), out.size, int(threads))

    def fnv(self, data) -> int:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        return int(self.L.tsqo_fnv1a64(_ptr(a), a.size))


class _RefCtx(C.Structure):
    # struct TSQCompressionContext { uint16_t *refhash; }  (turbosqueeze.h:57-63)
    _fields_ = [("refhash", C.c_void_p)]


class Reference:
    """The reference's own tsqEncode/tsqDecode (oracle/_ref/libtsq_ref.so), called under
    the canonical conditions of SURVEY.md 8c: zeroed table, zero-filled output, halo."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(os.path.join(HERE, "_ref", "libtsq_ref.so"))

    def __init__(self) -> None:
        L = C.CDLL(os.path.join(HERE, "_ref", "libtsq_ref.so"))
        # turbosqueeze.h:657,670
        L.tsqEncode.restype = None
        L.tsqEncode.argtypes = [C.POINTER(_RefCtx), _u8p, _u8p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        L.tsqDecode.restype = None
        L.tsqDecode.argtypes = [_u8p, _u8p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        self.L = L
        self._table = np.zeros(HASH_ENTRIES, dtype=np.uint16)
        self._ctx = _RefCtx(self._table.ctypes.data)

    def encode_block(self, data, ext: int, halo=None) -> bytes:
        n = len(data)
        buf = Oracle._with_halo(data, halo)
        out = np.zeros(n + (n >> 1) + (n >> 3) + 64, dtype=np.uint8)   # zero-filled (canonical)
        self._table[:] = 0                                            # tsqInit
        sz = C.c_uint32(0)
        self.L.tsqEncode(C.byref(self._ctx), _ptr(buf), _ptr(out), C.byref(sz), n, int(ext))
        return out[: sz.value].tobytes()

    def decode_block(self, stream, ext: int) -> bytes:
        s = np.zeros(len(stream) + 256, dtype=np.uint8)
        s[: len(stream)] = np.frombuffer(bytes(stream), dtype=np.uint8)
        out = np.zeros(BLOCK_SZ + 4096, dtype=np.uint8)
        sz = C.c_uint32(0)
        self.L.tsqDecode(_ptr(s), _ptr(out), C.byref(sz), len(stream), int(ext))
        return out[: sz.value].tobytes()
