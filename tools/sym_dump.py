#!/usr/bin/env python3
"""Debug: dump the symbol records block 0 of the instrumented orbit encoder produced."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_stats.so")
api._libs.clear()
L = api.lib()
data = np.fromfile(sys.argv[1], dtype=np.uint8)
lo, hi = int(sys.argv[2]), int(sys.argv[3])
lo_pos, hi_pos = int(sys.argv[5]), int(sys.argv[6])
codec = tsq.DeviceCodec(0)
blob = codec.compress(torch.from_numpy(data).cuda(), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
buf = (C.c_uint32 * 8192)()
L.tsqa_debug_syms(buf)
pos = 0
for k in range(hi):
    r = buf[k]
    if r >> 31:
        ln = ((r >> 22) & 15) + 1; st = r & 0x3FFFFF
        if k >= lo: print(k, f"LIT start={st} len={ln} (cum {pos})")
        pos += ln
    else:
        nib = (r >> 16) & 15; off = r & 0xFFFF
        if k >= lo: print(k, f"MATCH off={off} nib={nib} len={nib+1} (cum {pos})")
        pos += nib + 1

print("build log: base V SS nsym_entry lit_from after_match next_lane certain")
for k in range(400):
    d = buf[4096 + 10*k: 4096 + 10*k + 10]
    if d[1] == 0 and d[2] == 0: break
    V = d[1] | d[2] << 32; SS = d[3] | d[4] << 32; cm = d[8] | d[9] << 32
    if d[0] + 64 >= lo_pos and d[0] <= hi_pos:
        print(f"base={d[0]} nsym={d[5]} lit_from={d[6]} am={d[7]&1} next={d[7]>>8}")
        print("   V =", format(V, '064b')[::-1]); print("   SS=", format(SS, '064b')[::-1]); print("   cM=", format(cm, '064b')[::-1])
