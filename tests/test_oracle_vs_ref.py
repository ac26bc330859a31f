"""Differential test: oracle/tsq_oracle.c vs the reference's own tsq_encode.cpp/tsq_decode.cpp
(oracle/_ref/libtsq_ref.so).  Skipped where the reference build is absent.  CPU only."""
import os

import numpy as np
import pytest

import fuzzgen
import kat


@pytest.mark.parametrize("name", sorted(kat.KATS))
def test_kat_bytes_equal_reference(oracle, reference, name):
    data = bytes(kat.KATS[name][0]())
    for ext in (0, 1):
        a = oracle.encode_block(data, ext)
        b = reference.encode_block(data, ext)
        assert a == b
        assert reference.decode_block(a, ext) == data


def test_fuzz_small(oracle, reference):
    n_cases = int(os.environ.get("TSQ_FUZZ_CASES", "3000"))
    rng = np.random.default_rng(12345)
    for case in range(n_cases):
        n = int(rng.integers(1, 3000)) if case % 10 else int(rng.integers(1, 40))
        data = fuzzgen.structured(rng, n).tobytes()
        halo = rng.integers(0, 256, size=128, dtype=np.uint8).tobytes() if case % 3 else None
        if halo is not None and case % 2:
            halo = data[:128].ljust(128, b"\0")   # halo that continues the data: matches cross the end
        for ext in (0, 1):
            a = oracle.encode_block(data, ext, halo)
            b = reference.encode_block(data, ext, halo)
            assert a == b, (case, n, ext)
            back, st = oracle.decode_block(a, ext)
            assert st == 0 and back == data, (case, n, ext)
            assert reference.decode_block(a, ext) == data


def test_fuzz_large_blocks(oracle, reference):
    rng = np.random.default_rng(99)
    for case in range(6):
        n = int(rng.integers(70000, 1 << 21)) if case else 1 << 22
        data = fuzzgen.structured(rng, n).tobytes()
        halo = rng.integers(0, 256, size=128, dtype=np.uint8).tobytes()
        for ext in (0, 1):
            a = oracle.encode_block(data, ext, halo)
            assert a == reference.encode_block(data, ext, halo), (case, n, ext)
            back, st = oracle.decode_block(a, ext)
            assert st == 0 and back == data
