#!/usr/bin/env python3
"""Instrumented build (make stats): the dual WALK's waits, wavefront 0, block 0, per own tile.  Experiment tool."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ.get("STATS_LIB", "libturbosqueeze_amd_stats.so"))
api._libs.clear()
L = api.lib()
L.tsqa_debug_stats.argtypes = [C.c_void_p, C.c_void_p]
src = torch.from_numpy(tsq.synth.text(64 * (1 << 22), 1)).cuda()
codec = tsq.DeviceCodec(0)
blob = codec.compress(src, 0)
enc = (C.c_ulonglong * 64)()
L.tsqa_debug_stats(enc, None)
e = list(enc)
T = max(e[15], 1)
print(f"own tiles {T}; cycles per own tile: total {e[10]/T:.0f}  wait orbit {e[8]/T:.0f}  wait entry {e[11]/T:.0f} ({e[12]/T:.2f} polls)  wait confirm {e[29]/T:.0f}  wait answers {e[40]/T:.0f}  wait events {e[9]/T:.0f}")
print(f"entries already final {e[30]/T:.2f}  confirms {e[24]/T:.2f}  mispredictions {e[23]/T:.2f}  hazard lanes {e[26]/T:.2f}  queries {e[20]/T:.3f}")
