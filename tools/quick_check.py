#!/usr/bin/env python3
"""Fast development loop on the GPU box: bit-exactness of the device codec against the oracle on a handful of inputs
(both levels), then the kernel times of the headline job.  Experiment tool, not product.
  TSQ_LIB=libturbosqueeze_amd_x.so python tools/quick_check.py [--no-time] [--reps N]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
from oracle.pyoracle import Oracle

if os.environ.get("TSQ_LIB"):
    api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ["TSQ_LIB"])
    api._libs.clear()

orc = Oracle()
codec = tsq.DeviceCodec(0)
codec.set_variant(int(os.environ.get("ENC_VARIANT", "0")), int(os.environ.get("DEC_VARIANT", "0")))
B = 1 << 22
bad = 0


def check(name, host, ext):
    global bad
    src = torch.from_numpy(host).cuda()
    t0 = time.time()
    blob = codec.compress(src, ext)
    got = bytes(blob.cpu().numpy())
    want = orc.compress(host, ext, threads=8)
    ok = got == want
    back = codec.decompress(torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda())
    rt = torch.equal(back, src)
    if not ok:
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        print(f"  FAIL encode {name} ext={ext}: {len(got)} vs {len(want)} bytes, first difference at {k}")
    if not rt:
        print(f"  FAIL decode {name} ext={ext}")
    bad += (not ok) + (not rt)
    print(f"{name:28s} ext={ext} n={len(host):>10d} enc={'ok' if ok else 'BAD'} dec={'ok' if rt else 'BAD'} ({time.time() - t0:.1f}s)", flush=True)


import fuzzgen
rng = np.random.default_rng(7)
cases = [
    ("text 6 blocks + short", tsq.synth.text(6 * B + 300001, 3)),
    ("zeros 2 blocks", np.zeros(2 * B + 77, dtype=np.uint8)),
    ("random 2 blocks", tsq.synth.random_bytes(2 * B + 5, 5)),
    ("mix 3 blocks", tsq.synth.mix(3 * B + 12345, 9)),
    ("period64", np.tile(np.arange(64, dtype=np.uint8), B // 64 + 10)[: B + 500]),
    ("k4 pattern", ((np.arange(300000, dtype=np.int64) * 7 + 3) % 251).astype(np.uint8)),
    ("tiny 1", np.frombuffer(b"A", dtype=np.uint8).copy()),
    ("tiny 44", np.frombuffer(b"abcdefgh_abcdefgh_abcdefgh_XYZ_abcdefgh_abcd", dtype=np.uint8).copy()),
]
for s in range(3 if ("--enc-only" in sys.argv or "--dec-only" in sys.argv) else 12):
    cases.append((f"fuzz {s}", np.ascontiguousarray(fuzzgen.structured(np.random.default_rng(100 + s), int(rng.integers(1, 700000))))))
for name, host in cases:
    for ext in (0, 1):
        check(name, host, ext)
print("PARITY", "GREEN" if bad == 0 else f"RED ({bad})", flush=True)

if "--no-time" not in sys.argv:
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    host = tsq.synth.text(10 ** 9, 1)
    src = torch.from_numpy(host).cuda()
    out = torch.empty(api.container_bound(len(host)), dtype=torch.uint8, device="cuda")
    codec.profile(True)
    for ext in (((int(os.environ.get("QC_EXT", "0")),)) if ("--enc-only" in sys.argv or "--dec-only" in sys.argv) else (0, 1)):
        blob = codec.compress(src, ext, out)
        back = codec.decompress(blob)
        codec.profile_read()
        for _ in range(reps):
            blob = codec.compress(src, ext, out)
            back = codec.decompress(blob)
        torch.cuda.synchronize()
        em, en, dm, dn = codec.profile_read()
        assert torch.equal(back, src)
        print(f"TIME text 1e9 ext={ext}: encode kernel {em / max(en, 1):.2f} ms, decode kernel {dm / max(dn, 1):.3f} ms, ratio {blob.numel() / len(host):.4f}", flush=True)
if "--no-time" not in sys.argv and "--enc-only" not in sys.argv:
    # few blocks: one workgroup per block (variant 4), PARSE + COPY (6), two PARSE + COPY where the CUs allow (3)
    for nb in ((60,) if "--dec-only" in sys.argv else (30, 60, 120)):
        host = tsq.synth.text(nb * B, 2)
        src = torch.from_numpy(host).cuda()
        blob = codec.compress(src, 0)
        line = f"TIME decode {nb} blocks:"
        for dv in (4, 6, 3):
            codec.set_variant(0, dv)
            back = codec.decompress(blob)
            codec.profile_read()
            for _ in range(5):
                back = codec.decompress(blob)
            torch.cuda.synchronize()
            em, en, dm, dn = codec.profile_read()
            assert torch.equal(back, src), f"decode variant {dv} differs"
            line += f"  variant {dv}: {dm / max(dn, 1):.3f} ms ({len(host) / (dm / max(dn, 1)) / 1e6:.1f} GB/s)"
        codec.set_variant(0, 0)
        print(line, flush=True)
if "--big" in sys.argv:
    # the chip-filling regime: 4 GiB of text = 1 024 blocks (lean encoder layout, blocks queue behind each other)
    del src, out
    torch.cuda.empty_cache()
    host = tsq.synth.text(4 << 30, 1)
    src = torch.from_numpy(host).cuda()
    out = torch.empty(api.container_bound(len(host)), dtype=torch.uint8, device="cuda")
    for ext in (0, 1):
        blob = codec.compress(src, ext, out)
        back = codec.decompress(blob)
        codec.profile_read()
        for _ in range(2):
            blob = codec.compress(src, ext, out)
            back = codec.decompress(blob)
        torch.cuda.synchronize()
        em, en, dm, dn = codec.profile_read()
        assert torch.equal(back, src)
        print(f"BIG text 4 GiB ext={ext}: encode {len(host) / (em / max(en, 1)) / 1e6:.2f} GB/s ({em / max(en, 1):.2f} ms), decode {len(host) / (dm / max(dn, 1)) / 1e6:.1f} GB/s ({dm / max(dn, 1):.3f} ms)", flush=True)
sys.exit(1 if bad else 0)
