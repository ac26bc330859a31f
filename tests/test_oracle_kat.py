"""Pins oracle/tsq_oracle.c: (1) against the known-answer vectors of SURVEY.md 8c (sizes for
K0..K7, full byte strings for K0 and K1b), (2) against the committed golden fixtures that the
compiled reference produced (tests/golden/make_golden.py), (3) round trips.  CPU only."""
import json
import os

import numpy as np
import pytest

import kat

GOLDEN = kat.GOLDEN
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
SMALL = sorted(k for k in MANIFEST if os.path.exists(os.path.join(GOLDEN, k + ".in")))


def test_k0_exact_bytes(oracle):
    data = bytes(kat.KATS["K0"][0]())
    for ext in (0, 1):
        assert oracle.encode_block(data, ext).hex() == kat.K0_STREAM_HEX


def test_k1b_exact_bytes(oracle):
    for ext in (0, 1):
        assert oracle.encode_block(b"A", ext).hex() == kat.K1B_STREAM_HEX


@pytest.mark.parametrize("name", ["K0", "K1", "K2", "K4", "K5", "K7"])
def test_survey_sizes(oracle, name):
    make, n, sizes, _ = kat.KATS[name]
    data = bytes(make())
    assert len(data) == n
    for ext in (0, 1):
        out = oracle.encode_block(data, ext)
        assert len(out) == sizes[ext]
        back, st = oracle.decode_block(out, ext)
        assert st == 0 and back == data


@pytest.mark.parametrize("name", ["K3", "K6"])
def test_survey_sizes_full_block(oracle, name):
    make, n, sizes, _ = kat.KATS[name]
    data = bytes(make())
    for ext in (0, 1):
        out = oracle.encode_block(data, ext)
        assert len(out) == sizes[ext]
        assert "%016x" % oracle.fnv(out) == MANIFEST[name]["ext" if ext else "noext"]["fnv"]
        back, st = oracle.decode_block(out, ext)
        assert st == 0 and back == data


@pytest.mark.parametrize("name", ["K5", "K7"])
def test_manifest_hashes(oracle, name):
    data = bytes(kat.KATS[name][0]())
    assert "%016x" % oracle.fnv(data) == MANIFEST[name]["in_fnv"]
    for ext, tag in ((0, "noext"), (1, "ext")):
        assert "%016x" % oracle.fnv(oracle.encode_block(data, ext)) == MANIFEST[name][tag]["fnv"]


@pytest.mark.parametrize("name", SMALL)
def test_golden_fixture(oracle, name):
    data = open(os.path.join(GOLDEN, name + ".in"), "rb").read()
    for ext, tag in ((0, "noext"), (1, "ext")):
        want = open(os.path.join(GOLDEN, f"{name}.{tag}"), "rb").read()
        assert oracle.encode_block(data, ext) == want
        back, st = oracle.decode_block(want, ext)
        assert st == 0 and back == data


def test_reference_test_string_roundtrip(oracle):
    """What the reference's own test pins (test/test.cpp:30-54): ext=1 round trip of the 699-byte string."""
    data = bytes(kat.k1_input())
    out = oracle.encode_block(data, 1)
    back, st = oracle.decode_block(out, 1)
    assert st == 0 and back == data


def test_decode_rejects_garbage(oracle):
    assert oracle.decode_block(b"\xff\xff\xff\x00", 0) == (b"", 1)          # size header > 4 MiB
    assert oracle.decode_block(b"\x10\x00\x00\x00\x30\x05\x00", 0)[1] == 3  # match before block start
    good = oracle.encode_block(b"hello hello hello hello hello", 0)
    assert oracle.decode_block(good[:-3], 0)[1] == 2                        # truncated


def test_container_roundtrip_multiblock(oracle):
    rng = np.random.default_rng(7)
    import fuzzgen
    data = fuzzgen.structured(rng, (1 << 22) * 2 + 12345).tobytes()
    for ext in (0, 1):
        blob = oracle.compress(data, ext, threads=3)
        assert blob[:4] == b"TSQ1" and int.from_bytes(blob[4:8], "little") == 3
        assert int.from_bytes(blob[8:16], "little") == len(data)
        assert oracle.compress(data, ext, threads=1) == blob
        assert oracle.decompress(blob, threads=2) == data
