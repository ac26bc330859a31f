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


@pytest.mark.parametrize("inmem_max", ["", "1"])           # whole file in memory / streamed through pinned staging
def test_file_modes(tsq, oracle, tmp_path, inmem_max, monkeypatch):   # sample/main.cpp:144,160 use file mode
    if inmem_max:
        monkeypatch.setenv("TSQ_AMD_FILE_INMEM_MAX", inmem_max)
        monkeypatch.setenv("TSQ_AMD_FILE_BATCH_BLOCKS", "1")
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
