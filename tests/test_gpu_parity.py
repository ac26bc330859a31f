"""GPU parity tests: every call goes through the C ABI (libturbosqueeze_amd.so) and is compared
byte for byte with the oracle (oracle/tsq_oracle.c) and with the committed golden fixtures the
compiled reference produced.  Integer/byte work: the bar is bit-exact."""
import json
import os

import numpy as np
import pytest

import fuzzgen
import kat

pytestmark = pytest.mark.gpu

GOLDEN = kat.GOLDEN
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
SMALL = sorted(k for k in MANIFEST if os.path.exists(os.path.join(GOLDEN, k + ".in")))


@pytest.fixture(scope="module")
def tsq():
    import torch
    assert torch.cuda.is_available()
    import turbosqueeze_amd
    return turbosqueeze_amd


@pytest.fixture(scope="module")
def codec(tsq):
    c = tsq.DeviceCodec(0)
    yield c
    c.close()


def to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_bytes(t):
    return bytes(t.cpu().numpy())


# ---------------------------------------------------------------- single-block API (tsqEncode/tsqDecode)

@pytest.mark.parametrize("name", SMALL)
def test_golden_fixture_block_api(tsq, name):
    data = open(os.path.join(GOLDEN, name + ".in"), "rb").read()
    for ext, tag in ((0, "noext"), (1, "ext")):
        want = open(os.path.join(GOLDEN, f"{name}.{tag}"), "rb").read()
        assert tsq.tsq_encode(data, ext) == want
        assert tsq.tsq_decode(want, ext) == data


def test_k0_k1b_exact_bytes(tsq):
    for ext in (0, 1):
        assert tsq.tsq_encode(bytes(kat.KATS["K0"][0]()), ext).hex() == kat.K0_STREAM_HEX
        assert tsq.tsq_encode(b"A", ext).hex() == kat.K1B_STREAM_HEX


@pytest.mark.parametrize("name", sorted(kat.KATS))
def test_known_answer_vectors(tsq, oracle, name):
    make, n, sizes, _ = kat.KATS[name]
    data = bytes(make())
    for ext in (0, 1):
        out = tsq.tsq_encode(data, ext)
        assert len(out) == sizes[ext]
        assert out == oracle.encode_block(data, ext)
        if name in MANIFEST:
            assert "%016x" % oracle.fnv(out) == MANIFEST[name]["ext" if ext else "noext"]["fnv"]
        assert tsq.tsq_decode(out, ext) == data


def test_reference_test_scenario(tsq):
    """test/test.cpp:30-54: tsqEncode -> tsqDecode of the 699-byte string with extensions."""
    data = bytes(kat.k1_input())
    out = tsq.tsq_encode(data, 1)
    assert tsq.tsq_decode(out, 1) == data


def test_decode_rejects_bad_streams(tsq, oracle):
    assert tsq.tsq_decode(b"\xff\xff\xff\x00", 0) == b""                       # size header > 4 MiB
    assert tsq.tsq_decode(b"\x10\x00\x00\x00\x30\x05\x00", 0) == b""           # source before block start
    good = oracle.encode_block(b"hello hello hello hello hello", 0)
    assert tsq.tsq_decode(good[:-3], 0) == b""                                 # truncated


# ---------------------------------------------------------------- device-resident container path

def test_fuzz_containers_vs_oracle(codec, oracle):
    rng = np.random.default_rng(2024)
    for case in range(int(os.environ.get("TSQ_GPU_FUZZ", "120"))):
        n = int(rng.integers(1, 200000)) if case % 4 else int(rng.integers(1, 64))
        data = fuzzgen.structured(rng, n)
        for ext in (0, 1):
            blob = codec.compress(to_dev(data), ext)
            assert to_bytes(blob) == oracle.compress(data, ext), (case, n, ext)
            assert to_bytes(codec.decompress(blob)) == data.tobytes(), (case, n, ext)


def test_literal_runs_across_tiles_and_equal_byte_runs(codec, oracle, tsq):
    """Long literal runs whose 16-byte chunking is not aligned with the encoder's 64-position tiles
    (a full tile of literals with carried bytes), and runs of equal / short-period bytes (every
    lane shares its hash with its neighbours): the encoder's fast paths for both must stay exact."""
    rng = np.random.default_rng(77)
    cases = []
    for shift in range(0, 70, 7):
        head = (b"0123456789abcdef" * 8)[: 64 + shift]
        cases.append(np.frombuffer(head + rng.integers(0, 256, size=3000, dtype=np.uint8).tobytes() + head, dtype=np.uint8))
    for period in (1, 2, 3, 4, 5, 7, 16, 63, 64, 65):
        unit = rng.integers(0, 256, size=period, dtype=np.uint8)
        body = np.resize(unit, 5000)
        cases.append(np.concatenate([rng.integers(0, 256, size=37, dtype=np.uint8), body, rng.integers(0, 256, size=200, dtype=np.uint8), body[:777]]))
    cases.append(np.concatenate([np.zeros(70000, dtype=np.uint8), np.full(70000, 255, dtype=np.uint8), np.zeros(999, dtype=np.uint8)]))
    for k, data in enumerate(cases):
        for ext in (0, 1):
            blob = codec.compress(to_dev(data), ext)
            assert to_bytes(blob) == oracle.compress(data, ext), (k, ext)
            assert to_bytes(codec.decompress(blob)) == data.tobytes(), (k, ext)


def test_multiblock_halo_and_short_tail(codec, oracle, tsq):
    """Blocks are contiguous: block k's look-ahead reads block k+1 (SURVEY.md 8c canonical conditions)."""
    n = 3 * (1 << 22) + 77777
    host = tsq.synth.text(n, seed=11)
    host[(1 << 22) - 40:(1 << 22) + 40] = np.resize(np.frombuffer(b"crossing-the-block-edge ", dtype=np.uint8), 80)
    dev = to_dev(host)
    for ext in (0, 1):
        blob = codec.compress(dev, ext)
        want = oracle.compress(host, ext, threads=4)
        got = to_bytes(blob)
        assert got[:16] == want[:16]
        assert got == want
        import torch
        assert torch.equal(codec.decompress(blob), dev)


@pytest.mark.parametrize("kind", ["zeros", "random", "mix"])
def test_config5_inputs(codec, oracle, tsq, kind):
    n = 2 * (1 << 22) + 12345
    host = {"zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 5),
            "mix": lambda: tsq.synth.mix(n, 5)}[kind]()
    blob = codec.compress(to_dev(host), 1)
    assert to_bytes(blob) == oracle.compress(host, 1, threads=4)
    assert to_bytes(codec.decompress(blob)) == host.tobytes()


def test_enwik8_sized_bit_exact(codec, oracle, tsq):
    """BASELINE.json config 2: 100 MB, --no-ext, bit-exact vs the CPU path."""
    import torch
    n = 100_000_000
    host = tsq.synth.text(n, seed=8)
    dev = to_dev(host)
    blob = codec.compress(dev, 0)
    want = oracle.compress(host, 0, threads=os.cpu_count() or 8)
    assert blob.numel() == len(want)
    assert to_bytes(blob) == want
    assert torch.equal(codec.decompress(blob), dev)


def test_enwik9_sized_roundtrip_properties(codec, oracle, tsq):
    """BASELINE.json config 3 at full size: round trip identity, container structure, and a sampled
    block compared with the oracle (size-independent properties; full compare is config 2)."""
    import torch
    n = 1_000_000_000
    host = tsq.synth.text(n, seed=9)
    dev = to_dev(host)
    blob = codec.compress(dev, 0)
    head = to_bytes(blob[:16])
    assert head[:4] == b"TSQ1" and int.from_bytes(head[4:8], "little") == 239
    assert int.from_bytes(head[8:16], "little") == n
    ratio = blob.numel() / n
    assert 0.60 < ratio < 0.64, ratio
    back = codec.decompress(blob)
    assert torch.equal(back, dev)
    # walk the frames on the host and check three blocks against the oracle
    raw = blob.cpu().numpy()
    at, frames = 16, []
    for _ in range(239):
        ln = int(raw[at]) | int(raw[at + 1]) << 8 | (int(raw[at + 2]) & 0x7F) << 16
        frames.append((at + 3, ln))
        at += 3 + ln
    assert at == raw.size
    for b in (0, 117, 238):
        lo = b << 22
        hi = min(n, lo + (1 << 22))
        want = oracle.encode_block(host[lo:hi], 0, halo=bytes(host[hi:hi + 128]))
        s, ln = frames[b]
        assert bytes(raw[s:s + ln]) == want, b


def test_more_blocks_than_cus_lean_layouts(codec, oracle, tsq):
    """More blocks than CUs (1.25 GiB = 320 blocks, a 10 GiB / 8 GPU shard of BASELINE.json config 5): the default kernels
    switch to their lean layouts (two blocks per CU).  Round trip identity, with extensions, on the 50 % mix, and sampled
    blocks against the oracle."""
    import torch
    n = 5 * (1 << 28) + 12345
    nb = (n + (1 << 22) - 1) >> 22
    host = tsq.synth.mix(n, seed=13)
    dev = to_dev(host)
    blob = codec.compress(dev, 1)
    back = codec.decompress(blob)
    assert torch.equal(back, dev)
    del back
    raw = blob.cpu().numpy()
    assert int.from_bytes(bytes(raw[4:8]), "little") == nb
    at, frames = 16, []
    for _ in range(nb):
        ln = int(raw[at]) | int(raw[at + 1]) << 8 | (int(raw[at + 2]) & 0x7F) << 16
        assert int(raw[at + 2]) >> 7 == 1                      # the extension flag of the frame
        frames.append((at + 3, ln))
        at += 3 + ln
    assert at == raw.size
    for b in (0, 200, nb - 1):
        lo = b << 22
        hi = min(n, lo + (1 << 22))
        want = oracle.encode_block(host[lo:hi], 1, halo=bytes(host[hi:hi + 128]))
        s0, ln = frames[b]
        assert bytes(raw[s0:s0 + ln]) == want, b


def test_container_errors(codec, tsq, oracle):
    good = oracle.compress(tsq.synth.text(300000, 2), 0)
    bad_magic = b"TSQ2" + good[4:]
    zero_blocks = good[:4] + (0).to_bytes(4, "little") + good[8:]
    truncated = good[: len(good) // 2]
    big_frame = good[:16] + b"\xff\xff\x7f" + good[19:]
    for blob in (bad_magic, zero_blocks, truncated, big_frame):
        with pytest.raises(tsq.TsqError):
            codec.decompress(to_dev(np.frombuffer(blob, dtype=np.uint8)), out_cap=400000)
    # corrupt an offset inside the stream: must be flagged, not crash
    arr = np.frombuffer(good, dtype=np.uint8).copy()
    arr[40:60] = 0xFF
    try:
        out = codec.decompress(to_dev(arr), out_cap=400000)
        assert out.numel() == 300000          # decoded to *something* of the right size, or raised
    except tsq.TsqError:
        pass
    assert to_bytes(codec.decompress(to_dev(np.frombuffer(good, dtype=np.uint8)))) == bytes(tsq.synth.text(300000, 2))


def test_corrupted_containers_agree_with_oracle(codec, tsq, oracle):
    """Hardened decode: random byte damage inside the frames.  The device decoder must come back
    (every loop of the kernel is bounded) and must agree with the oracle's validating decoder: reject what it
    rejects, and produce the same bytes where the damaged stream is still well formed."""
    rng = np.random.default_rng(4242)
    n_cases = int(os.environ.get("TSQ_GPU_CORRUPT_CASES", "150"))
    agree_ok = agree_bad = 0
    for case in range(n_cases):
        n = int(rng.integers(20_000, 400_000))
        ext = int(rng.integers(0, 2))
        good = np.frombuffer(oracle.compress(fuzzgen.structured(rng, n), ext), dtype=np.uint8).copy()
        for _ in range(int(rng.integers(1, 6))):
            at = int(rng.integers(16, good.size))
            good[at] = rng.integers(0, 256) if rng.random() < 0.7 else good[at] ^ (1 << int(rng.integers(0, 8)))
        want = oracle.decompress(good)
        try:
            got = to_bytes(codec.decompress(to_dev(good), out_cap=n + 4096))
        except tsq.TsqError:
            got = None
        if want is None:
            assert got is None, f"case {case}: the oracle rejects this stream, the device decoded {len(got)} bytes"
            agree_bad += 1
        else:
            assert got == want, f"case {case}: both decoders accept the damaged stream but disagree"
            agree_ok += 1
    assert agree_ok + agree_bad == n_cases


def test_all_kernel_variants_agree(tsq, oracle):
    """Every encoder / decoder generation kept for A/B produces the oracle's bytes."""
    c = tsq.DeviceCodec(0)
    host = np.concatenate([tsq.synth.text(6_000_000, seed=21), tsq.synth.mix(3_000_000, seed=22)])
    dev = to_dev(host)
    for ext in (0, 1):
        want = oracle.compress(host, ext, threads=4)
        for enc_variant, dec_variant in ((0, 0), (1, 1), (2, 2), (3, 0), (4, 0), (5, 0), (6, 6)):
            c.set_variant(enc_variant, dec_variant)
            blob = c.compress(dev, ext)
            assert to_bytes(blob) == want, (enc_variant, ext)
            assert to_bytes(c.decompress(blob)) == host.tobytes(), (dec_variant, ext)
    c.close()
