"""CPU-only: the C-ABI library loads and exports every symbol include/turbosqueeze_amd.h and
include/turbosqueeze.h declare; without a GPU every compute entry point fails loudly (no CPU
fallback, nothing routed through oracle/)."""
import os
import re
import subprocess

import pytest

import turbosqueeze_amd as tsq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("turbosqueeze_amd.h", "turbosqueeze.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(tsqa?_?[A-Za-z_0-9]*)\s*\(", src))
    return {n for n in names if n.startswith("tsq") and not n.endswith("_fn")}


def test_library_exports_every_declared_symbol():
    L = tsq.lib()
    missing = [n for n in sorted(declared_symbols()) if not hasattr(L, n)]
    assert not missing, missing
    assert len(declared_symbols()) >= 28


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.run(["ldd", tsq.lib_path()], capture_output=True, text=True).stdout
    assert "tsq_oracle" not in out and "tsq_ref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "turbosqueeze_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".c", ".cpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "pyoracle" not in text and "tsq_oracle" not in text and "oracle/" not in text, f


def test_sizes_helpers():
    L = tsq.lib()
    assert L.tsqa_block_count(1) == 1 and L.tsqa_block_count(1 << 22) == 1 and L.tsqa_block_count((1 << 22) + 1) == 2
    assert L.tsqa_block_count(10**9) == 239
    assert tsq.container_bound(10**9) == 16 + 239 * (3 + tsq.OUTPUT_SZ)


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(tsq.TsqError):
        tsq.DeviceCodec(0)
    with pytest.raises(tsq.TsqError):
        tsq.tsq_encode(b"some bytes to encode", 0)
    assert tsq.tsq_decode(b"\x05\x00\x00\xff\x40hello", 0) == b""
    with pytest.raises(tsq.TsqError):
        tsq.tsq_compress_mt(b"x" * 100, False)


@pytest.mark.skipif(not os.path.exists("/root/reference/test/test.cpp"), reason="no reference tree here (build container only)")
@pytest.mark.parametrize("source", ["test/test.cpp", "sample/main.cpp"])
def test_reference_programs_link_against_this_library(tmp_path, source):
    """Drop-in at the ABI: the reference's own test program and sample program, compiled from where they lie against
    the reference's own headers, LINK against libturbosqueeze_amd.so with no unresolved symbol.  (Nothing is copied;
    -D_MSC_VER -mbmi selects platform.h's second branch because this image has no <stdbit.h>, as in oracle/Makefile.
    The programs are not run here: there is no GPU in the build container.)"""
    obj, exe = tmp_path / "prog.o", tmp_path / "prog"
    subprocess.run(["g++", "-std=c++17", "-O1", "-D_MSC_VER=1900", "-mbmi", "-c", os.path.join("/root/reference", source), "-o", str(obj)],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run(["g++", "-o", str(exe), str(obj), "-L" + os.path.dirname(tsq.lib_path()), "-lturbosqueeze_amd",
                        "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,--no-undefined"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    undefined = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    wanted = {l.split()[-1] for l in undefined.splitlines() if " tsq" in l}
    assert wanted, "the program does not reference the library at all?"
    exported = subprocess.run(["nm", "-D", "--defined-only", tsq.lib_path()], capture_output=True, text=True).stdout
    have = {l.split()[-1] for l in exported.splitlines()}
    assert wanted <= have, sorted(wanted - have)


def test_pmc_summary_keys_by_kernel_and_launch_shape(tmp_path):
    """tools/pmc_summary.py (the source of bench.py's roofline.traffic): a kernel launched at two grid sizes in one profiled command
    gets one entry per launch shape ("<kernel>@<blocks>") beside the all-launch entry, and bench.py picks the timed job's shape."""
    import csv
    import importlib.util
    import json
    import sys
    d = tmp_path / "fetch" / "x"
    d.mkdir(parents=True)
    rows = [("tsq::dec_sym_kernel(unsigned char const*)", 239 * 1024, 1024, 100.0), ("tsq::dec_sym_kernel(unsigned char const*)", 1024 * 1024, 1024, 400.0),
            ("tsq::dec_sym_kernel(unsigned char const*)", 1024 * 1024, 1024, 420.0), ("void at::native::fill(int)", 256, 256, 5.0)]
    for sub in ("fetch", "write"):
        dd = tmp_path / sub / "x"
        dd.mkdir(parents=True, exist_ok=True)
        with open(dd / "p_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Grid_Size", "Workgroup_Size", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for name, grid, wg, val in rows:
                w.writerow([grid, wg, name, sub.upper() + "_SIZE", val])
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "s.json"
    mod.pmc(str(out), [f"fetch={tmp_path / 'fetch'}", f"write={tmp_path / 'write'}"])
    s = json.load(open(out))
    assert s["fetch"]["tsq::dec_sym_kernel@239"]["per_dispatch"] == 100.0
    assert s["fetch"]["tsq::dec_sym_kernel@1024"]["per_dispatch"] == 410.0 and s["fetch"]["tsq::dec_sym_kernel@1024"]["dispatches"] == 2
    assert s["fetch"]["tsq::dec_sym_kernel"]["dispatches"] == 3 and not any("fill" in k for k in s["fetch"])
    # bench.py takes the entry of the timed job's own launch shape, and only from a summary made from the running kernel sources
    sys.path.insert(0, ROOT)
    import bench
    import turbosqueeze_amd as tsq
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump(s, open(prof / "r99_pmc_fetch_write.json", "w"))
    old_root = bench.ROOT
    try:
        bench.ROOT = str(tmp_path)
        got, src = bench.pmc_traffic("dec_", 239, tsq.source_fingerprint())
        assert got == int((2 * 100.0 + 100.0) * 1024) and "@239" in src
        got, why = bench.pmc_traffic("dec_", 500, tsq.source_fingerprint())
        assert got is None and "500 blocks" in why
        got, why = bench.pmc_traffic("dec_", 239, "another-fingerprint")
        assert got is None and "other kernel sources" in why
    finally:
        bench.ROOT = old_root


def test_bench_gpus_n_refuses_a_node_with_fewer_gpus():
    """`python bench.py --gpus 2` typed plainly is a launcher of two ranks; on a node that shows fewer than two GPUs it must exit
    non-zero without printing a JSON line (a line that says n_gpus 1 for --gpus 2 must be impossible)."""
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TSQ_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "visible" in r.stderr


def test_product_library_carries_no_experiment_switch():
    """tsqa_build_info(): the product library was compiled without any timing-only (wrong streams on purpose) or instrumentation
    switch; a build that passes one without -DTSQ_EXPERIMENT does not compile (csrc/tsq_experiment.h)."""
    info = tsq.build_info()
    assert "timing_only=0" in info and "instrumented=0" in info and "ab_variants=0" in info and "[switches: none]" in info, info
    assert "timing_only=0" in tsq.build_info(ab=True) and "ab_variants=1" in tsq.build_info(ab=True)
    assert "TSQ_JITTER" in tsq.build_info(ab="jitter") and "timing_only=0" in tsq.build_info(ab="jitter")
    csrc = os.path.join(ROOT, "turbosqueeze_amd", "csrc")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-DTSQ_X_NOHAZ", os.path.join(csrc, "tsq_runtime.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "TSQ_EXPERIMENT" in r.stderr, r.stderr[-500:]
    # and the encoder source carries only the guarded timing-only sites
    src = open(os.path.join(csrc, "tsq_enc_stage.cuh")).read()
    assert sum("TSQ_X_" in ln for ln in src.splitlines()) <= 10
