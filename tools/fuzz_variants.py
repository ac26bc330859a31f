#!/usr/bin/env python3
"""Structured-random inputs (tests/fuzzgen.py) through the encoder's standard and lean layouts (variants 0 and 6), both levels, against the
oracle; the decoder round trip with it.  python tools/fuzz_variants.py [cases] [seed]   (GPU box; a soak, not part of the test suite)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import turbosqueeze_amd as tsq
import fuzzgen, pyoracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
orc = pyoracle.Oracle()
c = tsq.DeviceCodec(0)
bad = 0
for k in range(cases):
    n = int(rng.integers(1, 600000)) if k % 5 else int(rng.integers(1, 200))
    if k % 50 == 49: n = int(rng.integers(1 << 22, 3 << 22))              # a few multi-block ones (block edges, look-ahead)
    data = fuzzgen.structured(rng, n)
    dev = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    for ext in (0, 1):
        want = orc.compress(data, ext)
        for v in (0, 6):
            c.set_variant(v, 0)
            blob = c.compress(dev, ext)
            got = blob.cpu().numpy().tobytes()
            if got != want or c.decompress(blob).cpu().numpy().tobytes() != data.tobytes():
                bad += 1; print("MISMATCH case %d n=%d ext=%d variant=%d" % (k, n, ext, v))
print("%d cases x 2 levels x 2 layouts: %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
