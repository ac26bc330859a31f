#!/usr/bin/env python3
"""Instrumented build (make stats): on which XCD did every workgroup of the several-workgroups-per-block decoder really run?
dec_duo_kernel maps workgroup w to (xcd = w % 8, slot = w / 8) and puts a block's PARSE and COPY workgroups on the same w % 8, on
the assumption that the hardware deals workgroups to the XCDs round-robin; the hand-off records then stay in that XCD's L2.  This
tool reads XCC_ID back from every workgroup and reports how often the assumption held.  python tools/duo_xcd.py"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_stats.so")
api._libs.clear()
L = api.lib()
L.tsqa_debug_duo_xcc.argtypes = [C.c_void_p]
B = 1 << 22
codec = tsq.DeviceCodec(0)
for nb, np_ in ((30, 2), (60, 2), (85, 2), (120, 1), (128, 1)):
    host = tsq.synth.text(nb * B, 2)
    src = torch.from_numpy(host).cuda()
    blob = codec.compress(src, 0)
    for rep in range(3):
        back = codec.decompress(blob)
        torch.cuda.synchronize()
        assert torch.equal(back, src)
        x = np.zeros(2048, dtype=np.uint32)
        assert L.tsqa_debug_duo_xcc(x.ctypes.data) == 0
        per = np_ + 1
        groups = (nb + 7) // 8
        same = follows = total = 0
        for b in range(nb):
            xcd, g = b % 8, b // 8
            ws = [((g * per + r) * 8 + xcd) for r in range(per)]
            ids = [int(x[w]) & 15 for w in ws if w < 2048 and x[w] >> 31]
            if len(ids) != per:
                continue
            total += 1
            same += len(set(ids)) == 1
            follows += all(i == xcd for i in ids)
        print(f"{nb:4d} blocks, {per} workgroups per block, launch {rep}: {same}/{total} blocks have all their workgroups on one XCD; "
              f"{follows}/{total} on XCD (workgroup index % 8)", flush=True)
        x[:] = 0
