#!/usr/bin/env python3
"""Instrumented build (make stats): timeline of the WALK wavefronts over consecutive tiles of block 0 (cycles relative to the previous
tile's commit).  Experiment tool."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ.get("STATS_LIB", "libturbosqueeze_amd_stats.so"))
api._libs.clear()
L = api.lib()
L.tsqa_debug_trace.argtypes = [C.c_void_p]
codec = tsq.DeviceCodec(0)
src = torch.from_numpy(tsq.synth.text(64 * (1 << 22), 1)).cuda()
codec.compress(src, 0); codec.compress(src, 0)
torch.cuda.synchronize()
tr = np.zeros(4096, dtype=np.uint32)
assert L.tsqa_debug_trace(tr.ctypes.data) == 0
tr = tr.reshape(16, 256).astype(np.int64)
def rel(a, b): return int((a - b + (1 << 31)) % (1 << 32) - (1 << 31))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
print("tile  period | ORBIT pub  record  entry  PRED  walked  commit   (cycles relative to the commit of tile t-1)")
acc = np.zeros(7); cnt = 0
for t in range(4, 250):
    ref = tr[9, t - 1]
    vals = [rel(tr[9, t], ref), rel(tr[7, t], ref), rel(tr[8, t], ref), rel(tr[12, t], ref), rel(tr[13, t], ref), rel(tr[14, t], ref), rel(tr[9, t], ref)]
    acc += np.array(vals); cnt += 1
    if t < 4 + rows: print("%4d %7d | %8d %7d %6d %6d %6d %6d" % tuple([t] + vals))
print("mean %7d | %8d %7d %6d %6d %6d %6d" % tuple(acc / cnt))
