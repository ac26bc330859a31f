#!/usr/bin/env python3
"""Phase timing of the fast kernels with the instrumented library (make -C turbosqueeze_amd/csrc stats).
Block 0 publishes s_memtime deltas per phase; this prints them.  Experiment tool, not product."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api

api.lib_path = lambda: os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_stats.so")
api._lib = None
L = api.lib()
L.tsqa_debug_stats.argtypes = [C.c_void_p, C.c_void_p]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 239 * (1 << 22)
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
ext = int(sys.argv[3]) if len(sys.argv) > 3 else 0
host = {"text": tsq.synth.text, "random": tsq.synth.random_bytes, "mix": tsq.synth.mix}.get(kind, None)
host = host(n, 1) if host else np.zeros(n, dtype=np.uint8)
src = torch.from_numpy(host).cuda()
codec = tsq.DeviceCodec(0)
blob = codec.compress(src, ext)
back = codec.decompress(blob)
assert torch.equal(back, src)
enc = (C.c_ulonglong * 16)()
dec = (C.c_ulonglong * 16)()
L.tsqa_debug_stats(enc, dec)
e = list(enc); d = list(dec)
print(f"ENC tile pipeline block0 (cycles): FRONT wait-for-parser={e[0]} commit={e[1]} classify={e[2]} tiles={e[3]} | PARSER wait-for-front={e[4]} work={e[5]} tiles={e[6]} symbols={e[9]}")
if e[3] and e[6]:
    print(f"  per tile: front wait={e[0]/e[3]:.0f} commit={e[1]/e[3]:.0f} classify={e[2]/e[3]:.0f} | parser wait={e[4]/e[6]:.0f} record-load={e[7]/e[6]:.0f} orbit={e[8]/e[6]:.0f} account={e[10]/e[6]:.0f} send={e[11]/e[6]:.0f} serial={e[12]/e[6]:.0f} publish+rest={e[5]/e[6]:.0f}")
print(f"  P6b detail (wave 0): init scan={d[9]} barrier waits={d[10]} read phases={d[11]} write phases={d[15]}")
names = ["P0 stage", "P1 spec", "P2 dbl", "P3 chain", "P4 expand+scan", "P5 syms", "P6a scatter", "P6b jump", "P7 flush"]
dt = sum(d[:9])
print(f"DEC block0: total ticks={dt} chunks={d[12]} jump rounds={d[13]} groups={d[14]}  thread0: vector history loads={d[9]} bytewise history loads={d[10]} pending bytes at P6b start={d[11]} image bytes={d[15]}")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {d[k]:10d}  {100.0*d[k]/max(dt,1):5.1f}%  per chunk {d[k]/max(d[12],1):8.1f}")
