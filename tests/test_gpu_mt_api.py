"""GPU tests of the reference's scheduler API (tsqCompress_MT & friends) re-expressing the
scenarios of the reference's test/test.cpp:57-331 against the HIP stream scheduler."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tsq():
    import torch
    assert torch.cuda.is_available()
    import turbosqueeze_amd
    return turbosqueeze_amd


def test_context_alloc_free(tsq):                       # test/test.cpp:57-70,133-146
    L = tsq.lib()
    for _ in range(2):
        c = L.tsqAllocateContextCompression_MT(False)
        assert c
        assert C.cast(c, C.POINTER(C.c_uint32))[0] >= 1   # num_cores is the first field
        L.tsqDeallocateContextCompression_MT(c)
        d = L.tsqAllocateContextDecompression_MT(False)
        assert d
        L.tsqDeallocateContextDecompression_MT(d)


def test_compress_mt_roundtrip_and_oracle(tsq, oracle):  # test/test.cpp:149-199
    data = bytes(kat.k1_input())
    for ext in (False, True):
        blob = tsq.tsq_compress_mt(data, ext)
        assert blob == oracle.compress(data, int(ext))
        assert tsq.tsq_decompress_mt(blob) == data


def test_compress_mt_multibatch(tsq, oracle):
    """Several batches through both pipeline lanes, including the halo across a batch boundary."""
    os.environ["TSQ_AMD_BATCH_BLOCKS"] = "2"
    try:
        host = tsq.synth.text(5 * (1 << 22) + 4321, seed=31)
        host[2 * (1 << 22) - 30:2 * (1 << 22) + 30] = np.resize(np.frombuffer(b"batch-edge ", dtype=np.uint8), 60)
        data = host.tobytes()
        for ext in (False, True):
            blob = tsq.tsq_compress_mt(data, ext)
            assert blob == oracle.compress(host, int(ext), threads=4)
            assert tsq.tsq_decompress_mt(blob) == data
    finally:
        del os.environ["TSQ_AMD_BATCH_BLOCKS"]


def test_multi_device_scheduler_path(tsq, oracle, monkeypatch):
    """The scheduler's spread over several devices (TSQ_AMD_DEVICES), exercised on one GPU by listing it twice: a job
    that would fit one batch is cut into one slice of consecutive blocks per listed device, each slice with its
    look-ahead, gathered on the host in block order.  Also with small batches, so that several batches per device
    are in flight."""
    host = tsq.synth.text(7 * (1 << 22) + 31337, seed=32)
    for edge in (3, 4, 6):                                  # matches across every possible slice boundary
        host[edge * (1 << 22) - 30:edge * (1 << 22) + 30] = np.resize(np.frombuffer(b"slice-edge ", dtype=np.uint8), 60)
    data = host.tobytes()
    monkeypatch.setenv("TSQ_AMD_DEVICES", "0,0")
    for batch in (None, "3"):
        if batch:
            monkeypatch.setenv("TSQ_AMD_BATCH_BLOCKS", batch)
        for ext in (False, True):
            blob = tsq.tsq_compress_mt(data, ext)
            assert blob == oracle.compress(host, int(ext), threads=4), (batch, ext)
            assert tsq.tsq_decompress_mt(blob) == data, (batch, ext)


def test_config4_eight_way_block_sharding_on_one_gpu(tsq, oracle, monkeypatch):
    """BASELINE.json config 4 as far as one GPU can show it: the enwik9-sized job (239 blocks) cut by the scheduler into eight slices
    of consecutive blocks, one per listed device (the one GPU listed eight times), each with its look-ahead, gathered on the host in
    block order -- through the reference's own entry points.  The container is the oracle's; the round trip is exact."""
    n = 1_000_000_000
    host = tsq.synth.text(n, seed=9)
    data = host.tobytes()
    monkeypatch.setenv("TSQ_AMD_DEVICES", "0,0,0,0,0,0,0,0")
    blob = tsq.tsq_compress_mt(data, False)
    assert blob == oracle.compress(host, 0, threads=os.cpu_count() or 8)
    assert tsq.tsq_decompress_mt(blob) == data


def test_progress_fires_once_per_block_in_order(tsq):
    """tsq_threads.cpp:248-254,654-655: progress_cb once per written block, fractions k / n_blocks in order."""
    L = tsq.lib()
    nb = 21
    data = tsq.synth.text(nb * (1 << 22) - 1000, seed=33).tobytes()
    src = C.create_string_buffer(data, len(data))
    seen = []
    pcb = tsq.api.PROGRESS_FN(lambda jobid, frac, user: seen.append((jobid, frac)))
    done = []
    dcb = tsq.api.DONE_FN(lambda jobid, ok, user: done.append((jobid, ok, len(seen))))
    cctx = L.tsqAllocateContextCompression_MT(False)
    out, sz = C.c_void_p(), C.c_size_t(0)
    jid = L.tsqa_compress_async_cb(cctx, src, len(data), False, C.byref(out), C.byref(sz), False, False, 0,
                                   C.cast(dcb, C.c_void_p), C.cast(pcb, C.c_void_p), None)
    L.tsqDeallocateContextCompression_MT(cctx)
    assert done == [(jid, True, nb)]                         # every progress call came before the completion call
    assert [f for _, f in seen] == [(k + 1) / nb for k in range(nb)]
    blob = C.string_at(out, sz.value)
    seen.clear(); done.clear()
    dctx = L.tsqAllocateContextDecompression_MT(False)
    src2 = C.create_string_buffer(blob, len(blob))
    out2, sz2 = C.c_void_p(), C.c_size_t(0)
    jid2 = L.tsqa_decompress_async_cb(dctx, src2, len(blob), False, C.byref(out2), C.byref(sz2), False,
                                      C.cast(dcb, C.c_void_p), C.cast(pcb, C.c_void_p), None)
    L.tsqDeallocateContextDecompression_MT(dctx)
    assert done == [(jid2, True, nb)]
    assert [f for _, f in seen] == [(k + 1) / nb for k in range(nb)]
    assert C.string_at(out2, sz2.value) == data
    tsq.api._libc.free(out); tsq.api._libc.free(out2)


def test_st_file_api(tsq, oracle, tmp_path):
    """tsqCompress(FILE*, FILE*) / tsqDecompress(FILE*, FILE*) (turbosqueeze.cpp:48-95,98-147): file -> .tsq container ->
    file, streams supplied and kept open by the caller, `level` ignored, silent return on a bad container."""
    L = tsq.lib()
    libc = tsq.api._libc
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.ftell.restype = C.c_long
    libc.ftell.argtypes = [C.c_void_p]
    host = tsq.synth.text(2 * (1 << 22) + 4242, seed=34)
    src, mid, dst = tmp_path / "in.bin", tmp_path / "out.tsq", tmp_path / "back.bin"
    src.write_bytes(host.tobytes())
    for ext, level in ((False, 0), (True, 3)):
        fin, fout = libc.fopen(str(src).encode(), b"rb"), libc.fopen(str(mid).encode(), b"wb")
        L.tsqCompress(fin, fout, ext, level)
        assert libc.ftell(fout) > 16                         # the caller's stream is still open and positioned after the data
        libc.fclose(fin); libc.fclose(fout)
        assert mid.read_bytes() == oracle.compress(host, int(ext), threads=2)
        fin, fout = libc.fopen(str(mid).encode(), b"rb"), libc.fopen(str(dst).encode(), b"wb")
        L.tsqDecompress(fin, fout)
        libc.fclose(fin); libc.fclose(fout)
        assert dst.read_bytes() == host.tobytes()
    # bad magic / short header: nothing written, no crash (turbosqueeze.cpp:106-117)
    for bad in (b"NOPE" + bytes(60), b"TSQ1" + bytes(5)):
        mid.write_bytes(bad)
        fin, fout = libc.fopen(str(mid).encode(), b"rb"), libc.fopen(str(dst).encode(), b"wb")
        L.tsqDecompress(fin, fout)
        libc.fclose(fin); libc.fclose(fout)
        assert dst.read_bytes() == b""


def test_hostile_header_is_rejected_before_allocation(tsq):
    """A 22-byte container that claims 25 600 blocks / 100 GiB must fail at once instead of committing memory."""
    blob = b"TSQ1" + (25600).to_bytes(4, "little") + (100 << 30).to_bytes(8, "little") + b"\x03\x00\x00\x01\x00\x00"
    assert tsq.tsq_decompress_mt(blob) is None
    blob = b"TSQ1" + (1).to_bytes(4, "little") + (4 << 20).to_bytes(8, "little") + b"\x03\x00\x00\x00\x00\x40"
    assert tsq.tsq_decompress_mt(blob) is None              # 4 MiB claimed from a 3-byte stream


def test_bad_arguments(tsq):                              # tsq_threads.cpp:415-418
    L = tsq.lib()
    c = L.tsqAllocateContextCompression_MT(False)
    out, sz = C.c_void_p(), C.c_size_t(0)
    assert not L.tsqCompress_MT(c, None, 10, False, C.byref(out), C.byref(sz), False, False, 0)
    buf = C.create_string_buffer(b"abc", 3)
    assert not L.tsqCompress_MT(c, buf, 0, False, C.byref(out), C.byref(sz), False, False, 0)
    L.tsqDeallocateContextCompression_MT(c)
    assert tsq.tsq_decompress_mt(b"NOPE" + bytes(40)) is None      # bad magic
    assert tsq.tsq_decompress_mt(b"TSQ1" + bytes(12) + bytes(40)) is None   # n_blocks == 0


# whole file in memory / streamed through pinned staging; the output side: collected in memory (small file), the MAPPED file whose
# pages fallocate provides ahead of the batches (TSQ_AMD_FILE_MAP_MIN=1 makes a 12 MB file "large"), positional writes
# (TSQ_AMD_FILE_NO_MMAP)
@pytest.mark.parametrize("inmem_max,sink", [("", ""), ("1", ""), ("", "map"), ("1", "map"), ("", "pwrite"), ("1", "pwrite")])
def test_file_modes(tsq, oracle, tmp_path, inmem_max, sink, monkeypatch):   # sample/main.cpp:144,160 use file mode
    if inmem_max:
        monkeypatch.setenv("TSQ_AMD_FILE_INMEM_MAX", inmem_max)
        monkeypatch.setenv("TSQ_AMD_FILE_BATCH_BLOCKS", "1")
    if sink:
        monkeypatch.setenv("TSQ_AMD_FILE_MAP_MIN", "1")
    if sink == "pwrite":
        monkeypatch.setenv("TSQ_AMD_FILE_NO_MMAP", "1")
    L = tsq.lib()
    host = tsq.synth.text(3 * (1 << 22) + 99999, seed=41)
    src = tmp_path / "in.bin"; mid = tmp_path / "out.tsq"; dst = tmp_path / "back.bin"
    src.write_bytes(host.tobytes())
    c = L.tsqAllocateContextCompression_MT(False)
    outp = C.c_char_p(str(mid).encode()); outpp = C.cast(C.pointer(outp), C.POINTER(C.c_void_p))
    assert L.tsqCompress_MT(c, C.c_char_p(str(src).encode()), 0, True, outpp, None, True, True, 0)
    L.tsqDeallocateContextCompression_MT(c)
    assert mid.read_bytes() == oracle.compress(host, 1, threads=2)
    d = L.tsqAllocateContextDecompression_MT(False)
    outp2 = C.c_char_p(str(dst).encode()); outpp2 = C.cast(C.pointer(outp2), C.POINTER(C.c_void_p))
    assert L.tsqDecompress_MT(d, C.c_char_p(str(mid).encode()), 0, True, outpp2, None, True)
    L.tsqDeallocateContextDecompression_MT(d)
    assert dst.read_bytes() == host.tobytes()


def test_mapped_output_larger_than_one_allocation_step(tsq, oracle, tmp_path):
    """A file output above 64 MiB (the default TSQ_AMD_FILE_MAP_MIN) is written through the mapping of a file whose pages are
    allocated by fallocate in 64 MiB steps, a bounded distance ahead of the batches (tsq_compat.hip: Sink::ensure_allocated): the
    decompressed file of a 40-block container crosses two steps; nothing may be copied into a range that was not allocated, and
    the file ends at its exact length."""
    L = tsq.lib()
    host = tsq.synth.text(40 * (1 << 22) + 4321, seed=47)
    blob = oracle.compress(host, 0, threads=4)
    mid = tmp_path / "big.tsq"; dst = tmp_path / "big.bin"
    mid.write_bytes(blob)
    d = L.tsqAllocateContextDecompression_MT(False)
    outp = C.c_char_p(str(dst).encode()); outpp = C.cast(C.pointer(outp), C.POINTER(C.c_void_p))
    assert L.tsqDecompress_MT(d, C.c_char_p(str(mid).encode()), 0, True, outpp, None, True)
    L.tsqDeallocateContextDecompression_MT(d)
    assert os.path.getsize(dst) == host.size
    assert dst.read_bytes() == host.tobytes()


def test_encode_lookahead_state_is_reported(tsq, oracle):
    """tsqEncode takes the reference's look-ahead behind a block with process_vm_readv; a caller can ask whether that worked
    (include/turbosqueeze_amd.h: tsqa_encode_lookahead_state): 1 = read, 2 = refused by the system (zeros seen)."""
    L = tsq.lib()
    data = tsq.synth.text(70000, seed=48).tobytes()
    assert tsq.tsq_encode(data[:50000], 0) is not None
    assert L.tsqa_encode_lookahead_state() in (1, 2)


def test_async_chain_ordering(tsq, oracle):               # test/test.cpp:202-331
    """N async compress jobs; each completion callback (scheduler thread) chains a decompress job on
    the other context; deallocation waits for everything (tsq_context.cpp:150-155)."""
    L = tsq.lib()
    n_jobs = int(os.environ.get("TSQ_ASYNC_JOBS", "200"))
    data = bytes(kat.k1_input())
    cctx = L.tsqAllocateContextCompression_MT(False)
    dctx = L.tsqAllocateContextDecompression_MT(False)
    src = C.create_string_buffer(data, len(data))
    couts = [(C.c_void_p(), C.c_size_t(0)) for _ in range(n_jobs)]
    douts = [(C.c_void_p(), C.c_size_t(0)) for _ in range(n_jobs)]
    order, progress, results, lock = [], [], [], threading.Lock()
    keep = []

    def make_done(k):
        def dec_done(jobid, ok, user):
            with lock:
                results.append((k, ok))
        dcb = tsq.api.DONE_FN(dec_done)
        keep.append(dcb)

        def comp_done(jobid, ok, user):
            with lock:
                order.append(jobid)
            assert ok
            out, sz = couts[k]
            L.tsqa_decompress_async_cb(dctx, out, sz.value, False, C.byref(douts[k][0]), C.byref(douts[k][1]), False,
                                       C.cast(dcb, C.c_void_p), None, None)
        ccb = tsq.api.DONE_FN(comp_done)
        keep.append(ccb)
        return ccb

    def on_progress(jobid, frac, user):
        with lock:
            progress.append((jobid, frac))
    pcb = tsq.api.PROGRESS_FN(on_progress)

    ids = []
    for k in range(n_jobs):
        jid = L.tsqa_compress_async_cb(cctx, src, len(data), False, C.byref(couts[k][0]), C.byref(couts[k][1]), False,
                                       bool(k & 1), 0, C.cast(make_done(k), C.c_void_p), C.cast(pcb, C.c_void_p), None)
        ids.append(jid)
    L.tsqDeallocateContextCompression_MT(cctx)       # blocks until all compress jobs (and their callbacks) ran
    L.tsqDeallocateContextDecompression_MT(dctx)     # then all chained decompress jobs
    assert ids == list(range(1, n_jobs + 1))
    assert order == ids                              # FIFO
    assert len(results) == n_jobs and all(ok for _, ok in results)
    assert len(progress) == n_jobs and all(abs(f - 1.0) < 1e-12 for _, f in progress)
    for k in range(n_jobs):
        got = C.string_at(douts[k][0], douts[k][1].value)
        assert got == data
        blob = C.string_at(couts[k][0], couts[k][1].value)
        assert blob == oracle.compress(data, k & 1)
        tsq.api._libc.free(douts[k][0]); tsq.api._libc.free(couts[k][0])


def test_cli_tool(tsq, oracle, tmp_path):                  # sample/main.cpp: `tsq c|d|b`
    """tools/tsq_cli (C++ against include/turbosqueeze.h) compresses to the oracle's container,
    decompresses it back, and its benchmark mode verifies its own round trip."""
    import json
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tsq_cli")
    assert os.path.exists(cli), "run __graft_entry__.build() first"
    host = tsq.synth.text((1 << 22) + 777, seed=5)
    src, comp, back = tmp_path / "in.bin", tmp_path / "out.tsq", tmp_path / "back.bin"
    src.write_bytes(host.tobytes())
    for flag, ext in (([], 1), (["--no-ext"], 0)):
        subprocess.run([cli, "c", str(src), str(comp)] + flag, check=True, timeout=300)
        assert comp.read_bytes() == oracle.compress(host, ext)
        subprocess.run([cli, "d", str(comp), str(back)], check=True, timeout=300)
        assert back.read_bytes() == host.tobytes()
    r = subprocess.run([cli, "b", "--synthetic", str(3 * (1 << 22) + 5), "--reps", "1"], check=True,
                       timeout=300, capture_output=True, text=True)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["output_correct"] is True and line["input_bytes"] == 3 * (1 << 22) + 5


def test_decode_stall_is_retried_by_the_mt_api(tsq, oracle, monkeypatch):
    """tsqDecompress_MT and tsqDecode decode on several workgroups per block when a batch has few blocks; a workgroup that gives up
    waiting for a sibling (TSQA_ERR_STALL -- forced here with a wait limit of one poll) must not turn a good container into a
    failure: the scheduler decodes the batch again on one workgroup per block."""
    monkeypatch.setenv("TSQ_AMD_DECODE_WAIT_LIMIT", "1")
    host = tsq.synth.text(9 * (1 << 22) + 777, seed=93)
    data = host.tobytes()
    blob = oracle.compress(host, 0, threads=4)
    assert tsq.tsq_decompress_mt(blob) == data
    small = data[:300_000]
    assert tsq.tsq_decode(oracle.encode_block(small, 1), 1) == small
