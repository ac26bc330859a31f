#!/usr/bin/env python3
"""BASELINE.json config 5 shapes on one GPU: zeros / random / 50 % mix / text, with extensions:
ratio and device-resident encode / decode GB/s (kernel time from the library's HIP events)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
if os.environ.get("TSQ_LIB"):
    api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ["TSQ_LIB"])
    api._libs.clear()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
ext = int(sys.argv[2]) if len(sys.argv) > 2 else 1
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kinds = sys.argv[4].split(',') if len(sys.argv) > 4 else ("zeros", "random", "mix", "text")
codec = tsq.DeviceCodec(0)
dec_variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
codec.set_variant(variant, dec_variant)
for kind in kinds:
    host = {"zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 3),
            "mix": lambda: tsq.synth.mix(n, 3), "text": lambda: tsq.synth.text(n, 3)}[kind]()
    src = torch.from_numpy(host).cuda()
    out = torch.empty(tsq.container_bound(n), dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
    codec.compress_async(src, ext, out); codec.decompress_async(out, nb, back); torch.cuda.synchronize()
    codec.profile(True)
    for _ in range(2):
        codec.compress_async(src, ext, out); torch.cuda.synchronize()
        csz, st = codec.last_size_status(); assert st == 0
        codec.decompress_async(out, nb, back); torch.cuda.synchronize()
    em, en, dm, dn = codec.profile_read(); codec.profile(False)
    assert torch.equal(back, src)
    print(json.dumps({"input": kind, "bytes": n, "ext": ext, "encode_variant": variant, "decode_variant": dec_variant, "ratio": round(csz / n, 4),
                      "encode_GBps": round(n / (em / en * 1e-3) / 1e9, 2), "decode_GBps": round(n / (dm / dn * 1e-3) / 1e9, 2)}))
    del src, out, back
