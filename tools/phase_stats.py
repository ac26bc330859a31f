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

api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ.get("STATS_LIB", "libturbosqueeze_amd_stats.so"))
api._libs.clear()
L = api.lib()
L.tsqa_debug_stats.argtypes = [C.c_void_p, C.c_void_p]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 239 * (1 << 22)
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
ext = int(sys.argv[3]) if len(sys.argv) > 3 else 0
host = {"text": tsq.synth.text, "random": tsq.synth.random_bytes, "mix": tsq.synth.mix}.get(kind, None)
host = host(n, 1) if host else np.zeros(n, dtype=np.uint8)
src = torch.from_numpy(host).cuda()
codec = tsq.DeviceCodec(0)
codec.set_variant(int(os.environ.get("ENC_VARIANT", "0")), int(os.environ.get("DEC_VARIANT", "0")))
blob = codec.compress(src, ext)
back = codec.decompress(blob)
assert torch.equal(back, src)
enc = (C.c_ulonglong * 128)()
dec = (C.c_ulonglong * 16)()
L.tsqa_debug_stats(enc, dec)
e = list(enc); d = list(dec)
T = max(e[15], 1)
print(f"ENC staged pipeline block0 (tiles parsed={e[15]}, symbols={e[16]}); cycles per tile, busy = total - waited:")
print("  HASH    total=%.0f waited(ring)=%.0f busy=%.0f" % (e[1] / T, e[0] / T, (e[1] - e[0]) / T))
print("  TWINS   total=%.0f waited(hash)=%.0f busy=%.0f" % (e[49] / T, e[48] / T, (e[49] - e[48]) / T))
print("          lanes behind a fold collision per tile=%.2f, settle rounds per tile=%.2f" % (e[32] / T, e[33] / T))
print("  IN      total=%.0f waited(hash)=%.0f busy=%.0f" % (e[58] / T, e[57] / T, (e[58] - e[57]) / T))
print("  MATCH   total=%.0f waited(scan)=%.0f waited(parser)=%.0f busy=%.0f" % (e[4] / T, e[2] / T, e[3] / T, (e[4] - e[2] - e[3]) / T))
print("          length-extension rounds per tile=%.2f (from the window ring %.2f), tiles classified twice (even wave)=%.3f" % (e[36] / T, e[35] / T, e[50] / T))
print("  COMMIT  total=%.0f waited=%.0f busy=%.0f" % (e[46] / T, e[45] / T, (e[46] - e[45]) / T))
print("  ORBIT   total=%.0f waited=%.0f busy=%.0f" % (e[7] / T, e[6] / T, (e[7] - e[6]) / T))
print("  WALK    total=%.0f waited(orbit)=%.0f waited(events)=%.0f waited(answers)=%.0f busy=%.0f  queries per tile=%.3f" % (e[10] / T, e[8] / T, e[9] / T, e[40] / T, (e[10] - e[8] - e[9] - e[40]) / T, e[20] / T))
print("  ACCOUNT total=%.0f waited(events)=%.0f waited(queue)=%.0f busy=%.0f  tiles with events=%.3f" % (e[43] / T, e[41] / T, e[42] / T, (e[43] - e[41] - e[42]) / T, e[44] / T))

print("  BUILDER total=%.0f waited=%.0f (ring %.0f) busy=%.0f" % (e[18] / T, e[17] / T, e[37] / T, (e[18] - e[17]) / T))
print("  EMIT    total=%.0f waited=%.0f busy=%.0f" % (e[39] / T, e[38] / T, (e[39] - e[38]) / T))
print("  parser events per tile: stale-truncations=%.2f segments=%.2f hazard-lanes=%.2f (hard %.2f) replays=%.2f" % tuple(e[k] / T for k in (23, 24, 26, 27, 28)))
print("  hazard lanes per tile by candidate: twin in tile=%.3f in t-1=%.3f in t-2=%.3f | resolved as match=%.3f" % tuple(e[k] / T for k in (29, 30, 31, 21)))
print(f"  P6b detail (wave 0): init scan={d[9]} barrier waits={d[10]} read phases={d[11]} write phases={d[15]}")
names = ["P0 stage", "P1 spec", "P2 dbl", "P3 chain(+flush)", "P4 expand+scan", "P5 pairs+copy", "P5 retry rounds", "P6 jump", "P7/bookkeeping"]
dt = sum(d[:9])
print(f"DEC block0: total ticks={dt} chunks={d[12]} retry rounds={d[13]} jump rounds={d[14]}  thread0: vector history loads={d[9]} bytewise history loads={d[10]} pending bytes at P6b start={d[11]} image bytes={d[15]}")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {d[k]:10d}  {100.0*d[k]/max(dt,1):5.1f}%  per chunk {d[k]/max(d[12],1):8.1f}")

try:
    w = (C.c_ulonglong * 48)()
    L.tsqa_debug_dec_waves.argtypes = [C.c_void_p]
    if L.tsqa_debug_dec_waves(w) == 0:
        w = list(w); ch = max(d[12], 1)
        print("  pointer jumping per wavefront of block 0 (cycles / waiting bytes / loop iterations per chunk):")
        print("   " + "  ".join(f"w{k}: {w[3*k]/ch:.0f}/{w[3*k+1]/ch:.0f}/{w[3*k+2]/ch:.1f}" for k in range(16)))
        cyc = [w[3*k] / ch for k in range(16)]
        print(f"   mean {sum(cyc)/16:.0f}, max {max(cyc):.0f} cycles per chunk (the sum over chunks of the per-chunk maximum is what the barrier waits for)")
except Exception as e:
    print("  (no per-wavefront jump counters in this library)", e)
