import ctypes as C, os, sys
import numpy as np, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_stats.so")
api._libs.clear()
L = api.lib()
data = np.fromfile(sys.argv[1], dtype=np.uint8)
codec = tsq.DeviceCodec(0)
blob = codec.compress(torch.from_numpy(data).cuda(), 0)
buf = (C.c_uint32 * 8192)()
L.tsqa_debug_syms(buf)
for k in range(0, 80):
    print(k, hex(buf[k]), hex(buf[2048 + k]))
for k in range(0, 30):
    print("item", k, [hex(x) for x in buf[4096 + 4 * k: 4100 + 4 * k]])
