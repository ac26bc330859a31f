#!/usr/bin/env python3
"""Development helper: one input (kind, bytes, seed, ext) against the oracle, several runs; prints the first differing block."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
from oracle.pyoracle import Oracle
if os.environ.get("TSQ_LIB"):
    api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ["TSQ_LIB"])
    api._libs.clear()
kind, n, seed, ext = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 3
host = getattr(tsq.synth, kind)(n, seed)
want = Oracle().compress(host, ext, threads=8)
codec = tsq.DeviceCodec(0)
src = torch.from_numpy(host).cuda()
nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
for r in range(runs):
    got = bytes(codec.compress(src, ext).cpu().numpy())
    if got == want:
        print("run", r, "ok"); continue
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
    hdr_g = np.frombuffer(got[: 16 + 4 * nb + 64], dtype=np.uint8)
    print("run", r, "DIFF first byte", k, "sizes", len(got), len(want))
    print("  got  header", got[:48].hex())
    print("  want header", want[:48].hex())
