"""Known-answer inputs K0..K7 of SURVEY.md section 8c and the expected sizes / FNV-1a64
hashes captured there from the unmodified reference under canonical conditions."""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def xorshift32_bytes(n: int, seed: int = 2463534242) -> np.ndarray:
    """x^=x<<13; x^=x>>17; x^=x<<5; b = x>>24  (K5/K6) -- scalar recurrence, done in C
    (turbosqueeze_amd/csrc/tsq_synth.c: tsq_synth_xorshift32)."""
    from turbosqueeze_amd import synth
    return synth.xorshift32(n, seed)


def k7_textlike(n: int = 300000, seed: int = 88172645) -> np.ndarray:
    alphabet = b"etaoin shrdlu\n"
    x = seed & 0xFFFFFFFF
    buf = bytearray(n)
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        b = alphabet[(x >> 8) % 14]
        if i >= 64 and ((x >> 20) & 3):
            b = buf[i - 1 - ((x >> 12) & 63)]
        buf[i] = b
    return np.frombuffer(bytes(buf), dtype=np.uint8).copy()


def k4_pattern(n: int = 100000) -> np.ndarray:
    i = np.arange(n, dtype=np.int64)
    return ((i * 7 + 3) % 251).astype(np.uint8)


def k1_input() -> np.ndarray:
    """The 699-byte input the reference's own tests use (test/test.cpp:26), kept as a data fixture."""
    return np.fromfile(os.path.join(GOLDEN, "k1_input.bin"), dtype=np.uint8)


# name -> (input factory, in_bytes, (out_noext, out_ext), (fnv_noext, fnv_ext))
KATS = {
    "K0": (lambda: np.frombuffer(b"abcdefgh_abcdefgh_abcdefgh_XYZ_abcdefgh_abcd", dtype=np.uint8),
           44, (46, 46), (0x62cce3e11feeb497, 0x62cce3e11feeb497)),
    "K1": (k1_input, 699, (570, 570), (0xca67e4da3f6bd252, 0xca67e4da3f6bd252)),
    "K2": (lambda: np.zeros(4096, dtype=np.uint8), 4096, (4261, 4261),
           (0x5107270d3cbd4779, 0x5107270d3cbd4779)),
    "K3": (lambda: np.zeros(1 << 22, dtype=np.uint8), 1 << 22, (4358149, 4358149),
           (0xf56e314d0ffe6ee9, 0xf56e314d0ffe6ee9)),
    "K4": (k4_pattern, 100000, (16823, 4546), (0x08cae0732cfefafe, 0x77cde27431f320aa)),
    "K5": (lambda: xorshift32_bytes(65536), 65536, (68101, 68101),
           (0x037d812de0deadfe, 0x037d812de0deadfe)),
    "K6": (lambda: xorshift32_bytes(1 << 22), 1 << 22, (4358097, 4358097),
           (0x9c799914021cbaf0, 0x9c799914021cbaf0)),
    "K7": (k7_textlike, 300000, (240055, 240055), (0x912c8d142557857b, 0x912c8d142557857b)),
}

K0_STREAM_HEX = (
    "2c0000" "ef" "ff"
    + b"abcdefgh_abcdefg".hex() + b"h_abcdefgh_XYZ_a".hex()
    + "06" + b"b".hex() + "0c00"
    + "30" + b"abcd".hex()
)
K1B_STREAM_HEX = "010000ff0041"   # input b"A"
K1_INPUT_FNV = 0xed3a59a043202e6e
K7_INPUT_FNV = 0xd327530cdcbb905e
