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


def frames_of(raw, nb):
    at, frames = 16, []
    for _ in range(nb):
        ln = int(raw[at]) | int(raw[at + 1]) << 8 | (int(raw[at + 2]) & 0x7F) << 16
        frames.append((at + 3, ln, int(raw[at + 2]) >> 7))
        at += 3 + ln
    assert at == raw.size
    return frames


def assert_same_container(raw, want, nb):
    """Byte compare of two containers; on a mismatch, name the first block that differs."""
    want = np.frombuffer(want, dtype=np.uint8)
    if raw.size == want.size and np.array_equal(raw, want):
        return
    fa, fb = frames_of(raw, nb), frames_of(want, nb)
    for b, (x, y) in enumerate(zip(fa, fb)):
        assert x[1] == y[1], f"block {b}: stream sizes differ ({x[1]} vs {y[1]})"
        assert np.array_equal(raw[x[0]:x[0] + x[1]], want[y[0]:y[0] + y[1]]), f"block {b}: stream bytes differ"
    raise AssertionError("containers differ outside the frames")


@pytest.mark.parametrize("ext", [0, 1])
def test_enwik9_sized_full_compare(codec, oracle, tsq, ext):
    """BASELINE.json config 3 at full size (10^9 B, 239 blocks, both levels): the WHOLE container byte for byte
    against the oracle, and the round trip."""
    import torch
    n = 1_000_000_000
    host = tsq.synth.text(n, seed=9)
    dev = to_dev(host)
    blob = codec.compress(dev, ext)
    raw = blob.cpu().numpy()
    assert bytes(raw[:4]) == b"TSQ1" and int.from_bytes(bytes(raw[4:8]), "little") == 239
    assert int.from_bytes(bytes(raw[8:16]), "little") == n
    assert 0.60 < raw.size / n < 0.64
    want = oracle.compress(host, ext, threads=os.cpu_count() or 8)
    assert_same_container(raw, want, 239)
    del want
    back = codec.decompress(blob)
    assert torch.equal(back, dev)


@pytest.mark.parametrize("kind", ["text", "mix", "zeros", "random"])
def test_more_blocks_than_cus_lean_layouts_full_compare(codec, oracle, tsq, kind):
    """More blocks than CUs (1.25 GiB = 321 blocks, a 10 GiB / 8 GPU shard of BASELINE.json config 5): the default
    kernels switch to their lean layouts (two blocks per CU).  With extensions, on each config-5 input: the WHOLE
    container against the oracle, and the round trip."""
    import torch
    n = 5 * (1 << 28) + 12345
    nb = (n + (1 << 22) - 1) >> 22
    assert nb > 256
    host = {"text": lambda: tsq.synth.text(n, seed=14), "mix": lambda: tsq.synth.mix(n, seed=13),
            "zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 15)}[kind]()
    dev = to_dev(host)
    blob = codec.compress(dev, 1)
    raw = blob.cpu().numpy()
    assert int.from_bytes(bytes(raw[4:8]), "little") == nb
    assert all(e == 1 for _, _, e in frames_of(raw, nb))           # the extension flag of every frame
    want = oracle.compress(host, 1, threads=os.cpu_count() or 8)
    assert_same_container(raw, want, nb)
    del want, raw
    back = codec.decompress(blob)
    assert torch.equal(back, dev)


@pytest.mark.parametrize("kind", ["zeros", "random", "mix"])
def test_config5_full_size_10gib(codec, oracle, tsq, kind):
    """BASELINE.json config 5 at its full size, with extensions: 10 GiB = 2 560 blocks on one GPU (ten rounds of blocks; the 8-GPU
    form shards the same blocks b % 8).  The WHOLE container against the oracle, and the round trip."""
    import torch
    n = 10 << 30
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
    except OSError:
        lim = "max"
    if lim != "max" and int(lim) < 6 * n:
        pytest.skip("host memory limit too small for a 10 GiB input")
    if torch.cuda.get_device_properties(0).total_memory < 6 * n:
        pytest.skip("device memory too small for a 10 GiB input")
    host = {"zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 25),
            "mix": lambda: tsq.synth.mix(n, 23)}[kind]()
    dev = to_dev(host)
    blob = codec.compress(dev, 1)
    back = codec.decompress(blob)
    assert torch.equal(back, dev)
    del back, dev
    raw = blob.cpu().numpy()
    del blob
    torch.cuda.empty_cache()
    want = np.frombuffer(oracle.compress(host, 1, threads=os.cpu_count() or 8), dtype=np.uint8)
    assert raw.size == want.size and np.array_equal(raw, want)


def test_config1_single_256k_block(tsq, oracle):
    """BASELINE.json config 1: one 262 144-byte block of enwik-shaped text through tsqEncode / tsqDecode, --no-ext."""
    data = tsq.synth.text(262144, seed=1).tobytes()
    for ext in (0, 1):
        stream = tsq.tsq_encode(data, ext)
        assert stream == oracle.encode_block(data, ext)
        assert int.from_bytes(stream[:3], "little") == 262144
        assert tsq.tsq_decode(stream, ext) == data


def test_tsqencode_lookahead_matches_reference_contract(tsq, oracle):
    """The reference's encoder reads past inputBlock[inputSize-1] (tsq_encode.cpp:74,126) and its scheduler relies
    on it (tsq_threads.cpp:109).  tsqEncode here does the same: looping over the blocks of one contiguous buffer gives
    the container's streams (block k's look-ahead is block k+1), and a block followed by zeros gives the canonical
    last-block stream."""
    n = (1 << 22) + 70000
    host = tsq.synth.text(n, seed=77)
    # a 16-byte phrase early in the block, and again straddling the block edge: 6 bytes inside, 10 in the next block.
    # With the look-ahead the match at the edge is 16 long; with zeros behind the block it is 6.
    phrase = np.frombuffer(b"QRSTUVWX12345678", dtype=np.uint8)
    host[100000:100016] = phrase
    host[(1 << 22) - 6:(1 << 22) + 10] = phrase
    first, rest = host[:1 << 22].tobytes(), host[1 << 22:].tobytes()
    for ext in (0, 1):
        with_next = tsq.tsq_encode(first, ext, halo=rest[:128])
        alone = tsq.tsq_encode(first, ext)
        assert with_next == oracle.encode_block(first, ext, halo=rest[:128])
        assert alone == oracle.encode_block(first, ext)
        assert with_next != alone                                   # the look-ahead is really used
        blob = oracle.compress(host, ext, threads=2)
        ln = int.from_bytes(blob[16:19], "little") & 0x7FFFFF
        assert blob[19:19 + ln] == with_next


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
    """The product kernels (staged encoder / ring decoder, their lean layouts, the serial baselines) and, from the A/B
    library (make ab), every superseded generation kept for measurements produce the oracle's bytes."""
    host = np.concatenate([tsq.synth.text(6_000_000, seed=21), tsq.synth.mix(3_000_000, seed=22)])
    dev = to_dev(host)
    want = {ext: oracle.compress(host, ext, threads=4) for ext in (0, 1)}
    for ab, variants in ((False, ((0, 0), (1, 1), (6, 3), (7, 4), (0, 5), (0, 6))), (True, ((0, 0), (5, 0)))):
        assert os.path.exists(tsq.lib_path(ab)), "run __graft_entry__.build() first"
        c = tsq.DeviceCodec(0, ab=ab)
        for ext in (0, 1):
            for enc_variant, dec_variant in variants:
                c.set_variant(enc_variant, dec_variant)
                blob = c.compress(dev, ext)
                assert to_bytes(blob) == want[ext], (ab, enc_variant, ext)
                assert to_bytes(c.decompress(blob)) == host.tobytes(), (ab, dec_variant, ext)
        c.close()
    # the product library does not carry the superseded kernels
    c = tsq.DeviceCodec(0)
    c.set_variant(5, 0)
    with pytest.raises(tsq.TsqError):
        c.compress(dev, 0)
    c.close()


def test_encoder_handoffs_under_jitter(tsq, oracle):
    """The staged encoder's wavefronts hand records to each other through LDS counters and rely on the LDS executing a wavefront's
    operations in order (tsq_enc_stage.cuh: stage_publish).  The stress build delays every publication by a pseudo-random time that
    differs from block to block: 96 copies of one block (and of a second, incompressible-in-parts one) in one launch run the
    pipeline under 96 interleavings each; every stream must be the oracle's, both levels."""
    assert os.path.exists(tsq.lib_path("jitter")), "run __graft_entry__.build() first"
    c = tsq.DeviceCodec(0, ab="jitter")
    B = 1 << 22
    for seed, maker in ((41, tsq.synth.text), (42, tsq.synth.mix)):
        one = maker(B, seed=seed)
        host = np.tile(one, 96)
        dev = to_dev(host)
        for ext in (0, 1):
            want = oracle.compress(host, ext, threads=8)
            # the standard layout (MATCH and ORBIT on two pairs of wavefronts without extensions, fused into four with them) and the
            # lean one (variant 6: fused at both levels)
            for variant in (0, 6):
                c.set_variant(variant, 0)
                assert to_bytes(c.compress(dev, ext)) == want, (seed, ext, variant)
    c.close()


def test_sharded_blocks_api(tsq, oracle):
    """tsqa_encode_blocks_async / tsqa_decode_blocks_async / tsqa_frames_{to,from}_host_async through ShardedCodec:
    one job dealt over 3 'ranks' (three contexts on this GPU, one after the other), frames gathered in ONE host
    container -> the oracle's container; every rank decodes its own frames back."""
    import torch
    from turbosqueeze_amd import sharding
    n = 7 * (1 << 22) + 4567
    host = tsq.synth.text(n, seed=91)
    host[3 * (1 << 22) - 25:3 * (1 << 22) + 25] = np.resize(np.frombuffer(b"shard-edge", dtype=np.uint8), 50)
    world = 3
    for ext in (0, 1):
        want = oracle.compress(host, ext, threads=4)
        hc = sharding.HostContainer("tsq_test_shard_%d_%d" % (os.getpid(), ext), tsq.container_bound(n), create=True)
        hc.register()
        try:
            codecs, coders, lays = [], [], []
            for r in range(world):
                lay = sharding.ShardLayout(n, r, world)
                c = tsq.DeviceCodec(0)
                sc = sharding.ShardedCodec(lay, sharding.DeviceBlocks(c), hc, ext)
                # world > 1 without torch.distributed: hand every coder the full size table by running them in turn
                codecs.append(c); coders.append(sc); lays.append(lay)
            # encode every shard, collect sizes (what the all-gather does), then place the frames
            all_sizes = np.zeros(lays[0].nb, dtype=np.uint32)
            shards = []
            for r in range(world):
                d = to_dev(lays[r].pack_input(host))
                shards.append(d)
                coders[r].blocks.encode(d, lays[r].n_local, lays[r].stride, lays[r].last_len, ext)
                coders[r].blocks.sync()
                all_sizes[np.asarray(lays[r].blocks)] = coders[r].blocks.sizes_tensor().cpu().numpy().astype(np.uint32)[:lays[r].n_local]
            frame_at, total = sharding.frame_offsets(all_sizes)
            for r in range(world):                       # (rank 0 writes the header; every rank its own frames: tsqa_sharded_place_async)
                assert coders[r].blocks.place(all_sizes, lays[r], ext, hc) == total
                coders[r].blocks.sync()
            assert total == len(want)
            assert bytes(hc.array[:total]) == want
            for r in range(world):
                back = torch.empty(lays[r].shard_bytes, dtype=torch.uint8, device="cuda")
                assert coders[r].decompress(total, back) == n
                assert np.array_equal(back.cpu().numpy(), lays[r].expected_output(host))
            for c in codecs:
                c.close()
        finally:
            hc.close()


@pytest.mark.parametrize("n_blocks,ext", [(40, 1), (100, 0), (128, 1), (200, 0)])
def test_default_decoder_at_every_workgroup_split(tsq, oracle, codec, n_blocks, ext):
    """The decode kernel the library picks BY ITSELF changes with the block count of a launch (tsq_launch.cuh: several workgroups per
    block when the launch leaves CUs free).  One container per range -- 40, 100, 128 and 200 blocks -- compressed and decompressed
    with the default variant; the container must be the oracle's and the round trip exact.  (the two-workgroup
    range was reached only by forcing the variant on a 3-block input.)"""
    n = n_blocks * (1 << 22) - 12345
    host = tsq.synth.text(n, seed=70 + n_blocks) if n_blocks != 128 else tsq.synth.mix(n, seed=71)
    dev = to_dev(host)
    codec.set_variant(0, 0)
    blob = codec.compress(dev, ext)
    want = oracle.compress(host, ext, threads=8)
    assert to_bytes(blob) == want
    import torch
    back = codec.decompress(to_dev(np.frombuffer(want, dtype=np.uint8)))
    assert torch.equal(back, dev)


def test_sharded_fetch_decode_refuses_a_container_of_another_job(tsq, oracle):
    """tsqa_sharded_fetch_decode_async takes its block count from the container, which is not trusted: a container with more
    (shorter) blocks than the device buffers were sized for, or one longer than the host mapping, is refused with TSQA_ERR_FORMAT
    before any copy is enqueued."""
    import torch
    from turbosqueeze_amd import sharding
    B = 1 << 22
    small = tsq.synth.text(2 * B, seed=5)                                   # the job the buffers are sized for: two blocks
    other = np.concatenate([tsq.synth.text(100_000, seed=6 + k) for k in range(5)])
    # five short blocks in one container (the format allows any block size <= 4 MiB): built from the oracle's block streams
    streams = [oracle.encode_block(bytes(other[k * 100_000:(k + 1) * 100_000]), 0) for k in range(5)]
    blob = b"TSQ1" + (5).to_bytes(4, "little") + (500_000).to_bytes(8, "little") + b"".join(len(s).to_bytes(3, "little") + s for s in streams)
    hc = sharding.HostContainer("tsq_test_foreign_%d" % os.getpid(), tsq.container_bound(2 * B), create=True)
    hc.register()
    c = tsq.DeviceCodec(0)
    try:
        hc.array[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        lay = sharding.ShardLayout(2 * B, 0, 1)
        d_streams = torch.zeros(lay.n_local * tsq.OUTPUT_SZ, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(lay.shard_bytes, dtype=torch.uint8, device="cuda")
        with pytest.raises(tsq.TsqError) as e:
            c.sharded_fetch_decode_async(hc.ptr, len(blob), 0, 1, d_streams, d_out)
        assert e.value.code == 4, e.value                                    # TSQA_ERR_FORMAT
        torch.cuda.synchronize()
        assert int(d_streams.max()) == 0 and int(d_out.max()) == 0          # nothing was written
        # the same container with buffers that hold five blocks decodes
        d_streams = torch.zeros(5 * tsq.OUTPUT_SZ, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(5 * B, dtype=torch.uint8, device="cuda")
        assert c.sharded_fetch_decode_async(hc.ptr, len(blob), 0, 1, d_streams, d_out) == 500_000
        torch.cuda.synchronize()
        assert c.status() == 0
        for k in range(5):
            assert bytes(d_out[k * B:k * B + 100_000].cpu().numpy()) == bytes(other[k * 100_000:(k + 1) * 100_000])
    finally:
        c.close()
        hc.close()


def test_sharded_decode_stall_is_retried(tsq, oracle):
    """A GPU of a sharded job holds few blocks, so its decode always takes the several-workgroups-per-block kernels; with the wait
    limit at one poll they give up at once (TSQA_ERR_STALL).  ShardedCodec.decompress must then decode the frames, still on the
    device, again on one workgroup per block (tsqa_sharded_decode_again_async) instead of failing a good container."""
    import torch
    from turbosqueeze_amd import sharding
    B = 1 << 22
    n = 11 * B + 4567
    host = tsq.synth.text(n, seed=58)
    lay = sharding.ShardLayout(n, 0, 1)
    hc = sharding.HostContainer("tsq_test_stall_%d" % os.getpid(), tsq.container_bound(n), create=True)
    hc.register()
    c = tsq.DeviceCodec(0)
    try:
        blocks = sharding.DeviceBlocks(c)
        sc = sharding.ShardedCodec(lay, blocks, hc, 0)
        d_shard = torch.from_numpy(lay.pack_input(host)).cuda()
        size = sc.compress(d_shard)
        assert bytes(hc.array[:size]) == oracle.compress(host, 0, threads=4)
        d_back = torch.zeros(lay.shard_bytes, dtype=torch.uint8, device="cuda")
        c.set_decode_wait_limit(1)
        assert sc.decompress(size, d_back) == n
        assert getattr(blocks, "stall_retries", 0) == 1                      # the first attempt did give up, the second one decoded
        assert torch.equal(d_back, torch.from_numpy(lay.expected_output(host)).cuda())
        # the stream-ordered C call alone reports the code
        c.sharded_fetch_decode_async(hc.ptr, size, 0, 1, blocks.slots, d_back)
        torch.cuda.synchronize()
        assert c.status() == 7
        c.sharded_decode_again_async(blocks.slots, d_back)
        torch.cuda.synchronize()
        assert c.status() == 0
        # a retry is bound to the buffers of the call it repeats, and to that call being the context's last use of its frame
        # descriptors: other buffers, or a retry after an ordinary decompress on the same context, are refused
        other = torch.zeros_like(d_back)
        with pytest.raises(tsq.TsqError) as e:
            c.sharded_decode_again_async(blocks.slots, other)
        assert e.value.code == 3, e.value
        blob = c.compress(torch.from_numpy(host[:3 * B]).cuda(), 0)
        assert torch.equal(c.decompress(blob), torch.from_numpy(host[:3 * B]).cuda())
        with pytest.raises(tsq.TsqError) as e:
            c.sharded_decode_again_async(blocks.slots, d_back)
        assert e.value.code == 3, e.value
    finally:
        c.close()
        hc.close()


def test_multi_workgroup_decode_beside_a_long_encode(tsq, oracle):
    """The multi-workgroup decoders hand chunk records from a block's PARSE workgroup(s) to its COPY workgroup and so need them
    resident together (tsq_dec_duo.cuh).  Here a 30-block container is decoded on one context while a 2 GiB encode (512 blocks:
    several rounds of workgroups that fill every CU) runs on another context and stream: the decode's workgroups are dispatched
    piecemeal as CUs come free.  Bytes must be right and nothing may stall into the waits' bound."""
    import time
    import torch
    B = 1 << 22
    big = to_dev(tsq.synth.text(512 * B, seed=81))
    small_host = tsq.synth.text(30 * B, seed=82)
    small = to_dev(small_host)
    enc, dec = tsq.DeviceCodec(0), tsq.DeviceCodec(0)
    try:
        blob = dec.compress(small, 0).clone()
        big_out = torch.empty(tsq.container_bound(big.numel()), dtype=torch.uint8, device="cuda")
        enc.compress(big[:8 * B], 0)                                         # (scratch allocated, kernels loaded)
        s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for dv in (0, 5, 6):                                                 # default (three workgroups at 30 blocks), always three, always two
            dec.set_variant(0, dv)
            back = torch.empty(30 * B, dtype=torch.uint8, device="cuda")
            t0 = time.perf_counter()
            with torch.cuda.stream(s_enc):
                enc.compress_async(big, 0, big_out)
            with torch.cuda.stream(s_dec):
                for _ in range(4):                                           # several decodes while the encode is in flight
                    dec.decompress_async(blob, 30, back)
            s_dec.synchronize()
            t_dec = time.perf_counter() - t0
            _, status = dec.last_size_status()
            s_enc.synchronize()
            assert status == 0, (dv, status)
            assert torch.equal(back, small), dv
            assert t_dec < 3.0, f"variant {dv}: four 30-block decodes beside the encode took {t_dec:.2f} s"
    finally:
        enc.close()
        dec.close()


def _run_bench(argv, env_extra, timeout=1200):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_sharded_path_over_rccl_with_one_rank(tsq):
    """bench.py's N > 1 line -- process group on the nccl (= RCCL) backend, all-gather of the sizes on the GPU, one host container in
    /dev/shm registered with hipHostRegister, tsqa_sharded_place_async, tsqa_sharded_fetch_decode_async -- executed with a world of
    ONE rank (the box has one GPU); the host-gathered container is compared with the oracle's inside bench.py."""
    r, line = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "300000000"],
                         dict(TSQ_BENCH_FORCE_SHARDED="1", TSQ_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                              RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    g = line["config4_host_gather"]
    assert g["container_equals_oracle"] is True and g["collective_backend"] == "nccl" and g["scaling"] == "strong"
    assert g["rank0_step_breakdown_ms"]["size_gather"] > 0
    assert line["n_gpus"] == 1 and line["process_group_world_size"] == 1 and line["scaling"] == "weak"
    assert line["config"]["container_equals_oracle"] is True


def test_bench_gpus_2_typed_plainly_launches_two_ranks(tsq):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must start two ranks itself and say n_gpus 2 -- here
    with both ranks on the box's one GPU over gloo (TSQ_BENCH_SHARE_GPU / TSQ_BENCH_BACKEND: the dry run of a 1-GPU box).  The line's
    value is the weak-scaled one (two jobs), the config-4 section is the one job dealt b % 2 with its container equal to the
    oracle's (the flag used to be parsed and ignored)."""
    r, line = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "150000000"],
                         dict(TSQ_BENCH_SHARE_GPU="1", TSQ_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["process_group_world_size"] == 2 and len(line["devices"]) == 2
    assert sorted(d["rank"] for d in line["devices"]) == [0, 1]
    assert line["distinct_devices"] == 1                                    # both ranks sat on GPU 0, and the line says so
    assert line["scaling"] == "weak" and line["config"]["jobs"] == 2 and line["config"]["container_equals_oracle"] is True
    assert line["value"] > 0 and abs(line["value"] - 2 * 150000000 / (line["ms_per_step"] * 1e-3) / 1e9) < 0.01 * line["value"] + 1e-3
    g = line["config4_host_gather"]
    assert g["container_equals_oracle"] is True and g["scaling"] == "strong" and g["collective_backend"] == "gloo"


def test_bench_refuses_more_gpus_than_visible_and_a_world_that_is_not_gpus(tsq):
    """--gpus N on a node with fewer than N GPUs exits non-zero without printing a line; so does a launcher that started another
    number of ranks than --gpus says (a line whose n_gpus is not what was asked for must be impossible)."""
    import torch
    have = torch.cuda.device_count()
    r, line = _run_bench(["--gpus", str(have + 1), "--steps", "1", "--warmup", "0", "--size", "50000000"], {}, timeout=600)
    assert r.returncode != 0 and line is None, (r.returncode, r.stdout[-500:])
    assert "visible" in r.stderr
    r, line = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "50000000"],
                         dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537"), timeout=600)
    assert r.returncode != 0 and line is None, (r.returncode, r.stdout[-500:])


def test_config4_eight_ranks_dealt_b_mod_8_host_container_equals_oracle(tsq, oracle):
    """BASELINE config 4's own split at full size: the 10^9 B job, block b -> rank b % 8 (tsq_threads.cpp:71), through ShardedCodec --
    eight 'ranks' (eight contexts on this GPU, in turn), every frame placed by tsqa_sharded_place_async in ONE host container, which
    must be the oracle's container byte for byte; then every rank brings its frames back and decodes its 29 or 30 blocks (the
    several-workgroups-per-block decoders: a GPU of the sharded job holds few blocks)."""
    import torch
    from turbosqueeze_amd import sharding
    n, world, ext = 1_000_000_000, 8, 0
    host = tsq.synth.text(n, seed=1)
    want = oracle.compress(host, ext, threads=8)
    hc = sharding.HostContainer("tsq_test_cfg4_%d" % os.getpid(), tsq.container_bound(n), create=True)
    hc.register()
    codecs = []
    try:
        lays = [sharding.ShardLayout(n, r, world) for r in range(world)]
        assert sorted(b for l in lays for b in l.blocks) == list(range(239)) and all(l.blocks == list(range(l.rank, 239, 8)) for l in lays)
        coders = []
        all_sizes = np.zeros(lays[0].nb, dtype=np.uint32)
        for r in range(world):
            c = tsq.DeviceCodec(0)
            codecs.append(c)
            sc = sharding.ShardedCodec(lays[r], sharding.DeviceBlocks(c), hc, ext)
            coders.append(sc)
            d = to_dev(lays[r].pack_input(host))
            sc.blocks.encode(d, lays[r].n_local, lays[r].stride, lays[r].last_len, ext)
            sc.blocks.sync()
            all_sizes[np.asarray(lays[r].blocks)] = sc.blocks.sizes_tensor().cpu().numpy().astype(np.uint32)[:lays[r].n_local]
            del d
        frame_at, total = sharding.frame_offsets(all_sizes)
        for r in range(world):                            # rank 0 writes the header, every rank its own frames
            assert coders[r].blocks.place(all_sizes, lays[r], ext, hc) == total
            coders[r].blocks.sync()
        assert total == len(want)
        assert hc.array[:total].tobytes() == want
        for r in range(world):
            back = torch.empty(lays[r].shard_bytes, dtype=torch.uint8, device="cuda")
            assert coders[r].decompress(total, back) == n
            assert np.array_equal(back.cpu().numpy(), lays[r].expected_output(host)), r
            del back
    finally:
        for c in codecs:
            c.close()
        hc.close()


def test_decode_stall_is_not_a_stream_error_and_is_retried(tsq, oracle):
    """A decode on several workgroups per block waits for sibling workgroups; when one does not show up within the wait limit the
    kernel reports TSQA_ERR_STALL (7), not a malformed stream.  With the limit set to one poll the waits give up at
    once: the stream-ordered entry point must report 7, and the synchronous one must decode again on one workgroup per block by
    itself and return the right bytes."""
    import torch
    B = 1 << 22
    host = tsq.synth.text(12 * B + 12345, seed=91)
    src = to_dev(host)
    c = tsq.DeviceCodec(0)
    try:
        blob = c.compress(src, 0).clone()
        assert bytes(blob.cpu().numpy()) == oracle.compress(host, 0, threads=4)
        c.set_variant(0, 5)                                  # always three workgroups per block
        c.set_decode_wait_limit(1)
        back = torch.zeros(src.numel(), dtype=torch.uint8, device="cuda")
        c.decompress_async(blob, 13, back)
        torch.cuda.synchronize()
        _, status = c.last_size_status()
        assert status == 7, status                           # TSQA_ERR_STALL: nothing wrong with the container
        out = c.decompress(blob)                             # synchronous: decodes again on one workgroup per block
        assert torch.equal(out, src)
        c.set_decode_wait_limit(1 << 24)
        assert torch.equal(c.decompress(blob), src)          # and the three-workgroup decode itself is fine with the normal limit
    finally:
        c.close()
