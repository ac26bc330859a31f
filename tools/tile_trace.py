#!/usr/bin/env python3
"""Instrumented build (make stats): when each encoder stage handed over each of 256 consecutive tiles of block 0 (cycles, relative to
WALK's hand-over of the tile three back), and the averages.  Experiment tool.
  python tools/tile_trace.py [rows]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_stats.so")
api._libs.clear()
L = api.lib()
L.tsqa_debug_trace.argtypes = [C.c_void_p]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ext = int(sys.argv[2]) if len(sys.argv) > 2 else 0
codec = tsq.DeviceCodec(0)
src = torch.from_numpy(tsq.synth.text(10 ** 9, 1)).cuda()
out = torch.empty(api.container_bound(src.numel()), dtype=torch.uint8, device="cuda")
codec.compress(src, ext, out); codec.compress(src, ext, out)
torch.cuda.synchronize()
tr = np.zeros(4096, dtype=np.uint32)
assert L.tsqa_debug_trace(tr.ctypes.data) == 0
tr = tr.reshape(16, 256).astype(np.int64)
print("HW_ID of waves 0..15 (simd = bits 4..5):", " ".join("%d:simd%d" % (w, (int(tr[15, w]) >> 4) & 3) for w in range(16)))
names = ["HASH go", "HASH pub", "TWINS pub", "MATCH go", "MATCH ver", "MATCH pub", "COMMIT pub", "ORBIT pub", "WALK go", "WALK pub", "gather", "ORBIT go", "MATCH pre", "MATCH lds"]
ix = {n: k for k, n in enumerate(names)}
W = tr[ix["WALK pub"]]
def rel(stage, t, ref_t):   # cycles from WALK's hand-over of tile ref_t to `stage` of tile t (mod 2^32)
    return int((tr[ix[stage], t] - W[ref_t] + (1 << 31)) % (1 << 32) - (1 << 31))
print("cycles relative to WALK pub of tile t-3 (F3); period = F(t)-F(t-1)")
hdr = ["tile", "period", "HASH go", "HASH pub", "TWINS pub", "MATCH go", "MATCH lds", "gather", "MATCH pre", "MATCH ver", "MATCH pub", "ORBIT go", "ORBIT pub", "WALK go", "WALK pub", "COMMIT(t-4)", "COMMIT(t-5)"]
print(" ".join("%11s" % h for h in hdr))
acc = np.zeros(len(hdr) - 1); cnt = 0
for t in range(8, 256):
    vals = [rel("WALK pub", t, t - 1)] + [rel(sname, t, t - 3) for sname in ("HASH go", "HASH pub", "TWINS pub", "MATCH go", "MATCH lds", "gather", "MATCH pre", "MATCH ver", "MATCH pub", "ORBIT go", "ORBIT pub", "WALK go", "WALK pub")] + [rel("COMMIT pub", t - 4, t - 3), rel("COMMIT pub", t - 5, t - 3)]
    acc += np.array(vals); cnt += 1
    if t < 8 + rows:
        print("%11d " % t + " ".join("%11d" % v for v in vals))
print("%11s " % "mean" + " ".join("%11d" % v for v in acc / cnt))
