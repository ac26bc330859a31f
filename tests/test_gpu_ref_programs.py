"""The reference's OWN programs, executed against libturbosqueeze_amd.so on the GPU (the drop-in boundary).

oracle/Makefile (target ref_programs) compiles /root/reference/test/test.cpp and /root/reference/sample/main.cpp, unmodified,
from where they lie, against the reference's own headers, and links them against this repo's library; the two binaries land in
oracle/_ref/ (git-ignored, they travel to the GPU box like oracle/_ref/libtsq_ref.so).  Here they run:
  * all ten argv-dispatched tests of test/test.cpp:334-362 must return 0;
  * `tsq c` / `tsq d` (sample/main.cpp:117-170, file -> file through tsqCompress_MT / tsqDecompress_MT) on a synthetic file:
    the .tsq container must be the oracle's, byte for byte, both levels, and the round trip must give the file back;
  * `tsq b` (sample/main.cpp:43-114) on an ./enwik9 of synthetic text must run to the end.
Skipped when the binaries are absent (no reference tree where the repo was built)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TEST = os.path.join(ROOT, "oracle", "_ref", "ref_test")
REF_TSQ = os.path.join(ROOT, "oracle", "_ref", "ref_tsq")

pytestmark = pytest.mark.gpu

# test/test.cpp:334-362
REFERENCE_TESTS = ["test_tsq_context", "test_tsq_compress", "test_tsq_context_mt", "test_tsq_compress_mt", "test_tsq_queue_mt",
                   "test_tsq_context_mt2", "test_tsq_decompress_mt", "test_tsq_compress_async_mt", "test_tsq_decompress_async_mt",
                   "test_tsq_massive_async_mt"]


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="oracle/_ref/ref_test not built (make -C oracle ref_programs needs /root/reference)")
@pytest.mark.parametrize("name", REFERENCE_TESTS)
def test_reference_test_program(name):
    r = subprocess.run([REF_TEST, name], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{name}: exit {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}"


@pytest.mark.skipif(not os.path.exists(REF_TSQ), reason="oracle/_ref/ref_tsq not built (make -C oracle ref_programs needs /root/reference)")
@pytest.mark.parametrize("no_ext", [True, False])
def test_reference_sample_program_files(tmp_path, oracle, no_ext):
    import turbosqueeze_amd as tsq
    host = np.concatenate([tsq.synth.text(9_500_000, seed=31), tsq.synth.mix(3_100_003, seed=32)])     # four blocks, the last one short
    src, packed, back = tmp_path / "input.bin", tmp_path / "input.tsq", tmp_path / "input.out"
    host.tofile(src)
    cmd = [REF_TSQ, "c", str(src), str(packed)] + (["--no-ext"] if no_ext else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    want = oracle.compress(host, 0 if no_ext else 1, threads=4)
    got = packed.read_bytes()
    assert got == want, f"container differs from the oracle's: {len(got)} vs {len(want)} bytes"
    r = subprocess.run([REF_TSQ, "d", str(packed), str(back)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert back.read_bytes() == host.tobytes()


@pytest.mark.skipif(not os.path.exists(REF_TSQ), reason="oracle/_ref/ref_tsq not built")
def test_reference_sample_program_benchmark(tmp_path):
    """`tsq b` reads ./enwik9, compresses and decompresses it memory to memory through the _MT API and prints its rates."""
    import turbosqueeze_amd as tsq
    tsq.synth.text(50_000_000, seed=33).tofile(tmp_path / "enwik9")
    r = subprocess.run([REF_TSQ, "b"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "output_correct: 1" in r.stdout and "MB/s" in r.stdout, r.stdout[-1500:]      # sample/main.cpp:99-113
